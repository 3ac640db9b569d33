"""Detection log: the ``*_dcnn_data.csv`` file boundary.

Producer side follows /root/reference/dcnn/scripts/tests/visualize_uav.py:117-141
(``generate_log_oneline``) and :223-233 (file layout).  The files the reference *ships*
(data/static_dcnn_data.csv, data/dynamic_dcnn_data.csv) and its consumer
(/root/reference/aruco_detect.py:105-123 ``readCentroidData``: skips 2 lines, reads exactly 17
columns with ``int(cell)``, ''/'nan' -> 0) need a different, post-edited shape:
``Host id: <id>`` + 16 commas, integer cells, columns ordered host first then the vehicles that
map to ArUco ids 1..3.  ``write_consumer_csv`` emits that shape (default), ``write_raw_csv`` the
literal script output.
"""
import math

import numpy as np


def generate_log_oneline(objects, host_id, frame_idx, closest_lookup=None):
    """objects: ObjectInstances of this frame (``next_frame`` result).  Returns (line, highest_id).

    ``closest_lookup(obj_index, host_index)`` may supply the closest point from the GPU's
    per-frame table; otherwise ``utils.mask_utils`` is called like the reference does."""
    from . import mask_utils
    if len(objects) == 0:
        return "", 0
    centroids = [mask_utils.get_mask_centroid(m) for m in objects.pred_masks]
    ids = list(objects.ids)
    if host_id in ids:
        hidx = ids.index(host_id)
        hc = centroids[hidx]
        if closest_lookup is not None:
            closest = [closest_lookup(k, hidx) for k in range(len(ids))]
        else:
            closest = [mask_utils.compute_closest_point(m, hc) for m in objects.pred_masks]
    else:
        closest = [("nan", "nan")] * len(ids)
    result = [str(frame_idx)]
    highest = max(ids)
    for ob_id in range(1, highest + 1):
        if ob_id in ids:
            k = ids.index(ob_id)
            result += [str(centroids[k][0]), str(centroids[k][1]), str(closest[k][0]), str(closest[k][1])]
        else:
            result += [""] * 4
    return ",".join(result), highest


def write_raw_csv(path, lines, host_id, max_obj_id):
    header = "frame"
    for i in range(1, max_obj_id + 1):
        header += ",id_{} cent_x,id_{} cent_y,id_{} clos_x,id_{} clos_y".format(i, i, i, i)
    with open(path, "w") as f:
        f.write("Ford id: {}\n".format(host_id))
        f.write(header + "\n")
        f.write("\n".join(lines))


def _cell_to_int(cell):
    if cell == "" or cell == "nan":
        return ""
    v = float(cell)
    if math.isnan(v):
        return ""
    return str(int(v))


def consumer_rows(lines, host_id, vehicle_ids):
    """Re-orders raw lines into the shipped 17-column integer layout: host, then vehicle_ids (3)."""
    order = [host_id] + list(vehicle_ids)
    rows = []
    for line in lines:
        cells = line.split(",") if line else [""]
        out = [cells[0] if cells[0] != "" else ""]
        for oid in order:
            base = 1 + 4 * (oid - 1)
            seg = cells[base:base + 4] if len(cells) >= base + 4 else [""] * 4
            out += [_cell_to_int(v) for v in seg]
        rows.append(out)
    return rows


def write_consumer_csv(path, lines, host_id, vehicle_ids, first_frame=0):
    """17 columns, ints, blanks for absent objects, '\\n' endings and a trailing newline."""
    order = [host_id] + list(vehicle_ids)
    assert len(order) == 4, "the consumer reads the host + 3 vehicles (aruco_detect.py:634-720)"
    header = "frame" + "".join(",id_{0} cent_x,id_{0} cent_y,id_{0} clos_x,id_{0} clos_y".format(i) for i in order)
    rows = consumer_rows(lines, host_id, vehicle_ids)
    with open(path, "w") as f:
        f.write("Host id: {}".format(host_id) + "," * 16 + "\n")
        f.write(header + "\n")
        for k, r in enumerate(rows):
            if r[0] == "":
                r[0] = str(first_frame + k)
            f.write(",".join(r) + "\n")


def read_centroid_data(path):
    """The consumer's parser, restated from aruco_detect.py:105-123 (used by the round-trip tests)."""
    import csv
    data = []
    with open(path) as f:
        for n, row in enumerate(csv.reader(f, delimiter=",")):
            if n > 1:
                data.append([0 if (row[i] == "" or row[i] == "nan") else int(row[i]) for i in range(17)])
    return data


def pixel_distance(a, b):
    """aruco_detect.py:483-492 pixel part of calculateDistance."""
    return float(np.sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1])))


def calculate_distance(lidar, aruco, bbox, marker_length, msp4, msp):
    """The consumer's only arithmetic on the CSV values, restated from
    /root/reference/aruco_detect.py:483-492 ``calculateDistance``: pixel distances from the host's lidar
    point to the vehicle's marker and to its closest point, converted to metres with the mean marker size
    in pixels.  Returns (dist_aruco, dist_bbox)."""
    d_aruco = pixel_distance(lidar[0], aruco[0])
    d_bbox = pixel_distance(lidar[0], bbox[0])
    scale = marker_length / ((msp4 + msp) / 2)
    return d_aruco * scale, d_bbox * scale


def dcnn_points(row, vehicle):
    """Columns of one parsed CSV row the consumer uses (aruco_detect.py:634,665-666,692-693,719-720):
    vehicle 0 = host centroid [1:3]; vehicles 1..3 = (centroid [5+4k:7+4k], closest point [7+4k:9+4k])."""
    if vehicle == 0:
        return (row[1], row[2]), None
    b = 5 + 4 * (vehicle - 1)
    return (row[b], row[b + 1]), (row[b + 2], row[b + 3])
