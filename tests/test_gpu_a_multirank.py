"""-m gpu: the frame-sharded driver with TWO ranks on the one GPU of the test box (gloo rendezvous / gather, both ranks
computing on cuda:0 with the real HIP path), and the resumed-run switch of the same driver.

The sharded loop is /root/reference/dcnn/scripts/tests/visualize_uav.py:186-221 (frames are independent until the
association); `--start-frame` is START_FROM_FRAME of visualize_uav.py:172-190 (earlier frames are read and dropped, the
tracker's first frame is frame S, the log keeps absolute frame numbers).

Every run here is a fresh child process of `tools/run_sequence.py` (which starts its own rank processes for --gpus 2):
this file sorts first among the -m gpu files so the pytest process has not touched the GPU when the children start, and
it never does in this file.  RCCL refuses two ranks on one device, hence APSE_DIST_BACKEND=gloo for the rehearsal; the
"nccl" branch is the same code with device tensors (tests/test_gpu_rccl.py runs it with one rank).
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tools", "run_sequence.py")


def _run(args, tmp_path, name, env_extra=None, timeout=900):
    out = str(tmp_path / (name + ".csv"))
    raw = str(tmp_path / (name + "_raw.csv"))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, SCRIPT] + args + ["--out", out, "--raw-out", raw], env=env, cwd=ROOT, timeout=timeout,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout[-4000:]
    with open(out) as f, open(raw) as g:
        return f.read(), g.read()


def test_two_ranks_on_one_gpu_equal_one_rank(tmp_path, logdir):
    """`tools/run_sequence.py --gpus 2` (ranks started by the tool, contiguous shards, one gather, rank-0 C++ replay) must
    write the CSV of the one-rank run byte for byte -- consumer layout and the script-literal raw layout."""
    common = ["--frames", "7", "--kind", "dynamic", "--size", "2160x3840"]        # uneven shards: 4 + 3 frames
    one, one_raw = _run(common, tmp_path, "one")
    two, two_raw = _run(common + ["--gpus", "2"], tmp_path, "two", {"APSE_DIST_BACKEND": "gloo"})
    with open(os.path.join(logdir, "multirank.log"), "a") as f:
        f.write("one-rank csv (%d bytes):\n%s\ntwo-rank csv equal: %s, raw equal: %s\n" % (len(one), one[:600], one == two, one_raw == two_raw))
    assert len(one.split("\n")) == 2 + 7 + 1                   # two header lines, 7 rows, trailing newline
    assert any(c for c in one.split("\n")[2].split(",")[1:])   # the rows carry detections
    assert two == one
    assert two_raw == one_raw


def test_config3_two_ranks_bf16_batch4_preproc_given_boxes(tmp_path, logdir):
    """BASELINE configs[3] as SURVEY 8d restates it (dynamic sequence, batch 4, bf16 storage / f32 accumulate, fused undistort +
    gamma, frame-sharded, gather -> single ordered CSV identical to the config-2 CSV), in the deterministic given-boxes form
    (track_rcnn.py:52-54): 18 frames as 10 + 8 on two ranks -- ragged last batches on both -- against the one-rank run."""
    common = ["--frames", "18", "--kind", "dynamic", "--size", "2160x3840", "--dtype", "bf16", "--batch", "4", "--preproc", "--given-boxes"]
    one, one_raw = _run(common, tmp_path, "c3_one")
    two, two_raw = _run(common + ["--gpus", "2"], tmp_path, "c3_two", {"APSE_DIST_BACKEND": "gloo"})
    rows = one.split("\n")
    with open(os.path.join(logdir, "multirank.log"), "a") as f:
        f.write("configs[3] given-boxes bf16 batch 4 preproc: two-rank csv equal: %s, raw equal: %s\n%s\n" % (one == two, one_raw == two_raw, one[:500]))
    assert len(rows) == 2 + 18 + 1
    assert all(sum(1 for c in r.split(",")[1:] if c) >= 8 for r in rows[2:-1])      # >= 2 vehicles with centroid + closest point per row
    assert two == one and two_raw == one_raw


def test_start_frame_rows_equal_full_run(tmp_path, logdir):
    """--start-frame 3 on a 9-frame sequence.  Static sequence (zero motion: every frame holds the same vehicles, so the
    tracker that starts at frame 3 issues the same ids in the same order): its rows must be rows 3.. of the full run,
    text for text, absolute frame numbers included."""
    common = ["--frames", "9", "--kind", "static", "--size", "2160x3840"]
    full, full_raw = _run(common, tmp_path, "full")
    part, part_raw = _run(common + ["--start-frame", "3"], tmp_path, "part")
    frows, prows = full.split("\n"), part.split("\n")
    assert frows[:2] == prows[:2]                              # header lines
    assert len(prows) == 2 + 6 + 1
    assert prows[2:] == frows[2 + 3:]
    assert [r.split(",")[0] for r in prows[2:-1]] == ["3", "4", "5", "6", "7", "8"]
    assert part_raw.split("\n")[2:] == full_raw.split("\n")[2 + 3:]
    with open(os.path.join(logdir, "multirank.log"), "a") as f:
        f.write("start-frame 3: rows %s == rows 3.. of the full run\n" % [r.split(",")[0] for r in prows[2:-1]])
