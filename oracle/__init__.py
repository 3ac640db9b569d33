"""CPU oracle for the apse_uav dcnn/ hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch-CPU / numpy restatement of the reference's
per-frame path (4K frame -> Mask R-CNN R-101-FPN -> embedding tracker ->
``*_dcnn_data.csv``).  It exists to *check* the HIP product path:

* only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
  ``cpu_baseline`` leg may import it;
* nothing under ``apse_uav_amd/`` imports it, and the product path never falls
  back to it (the product raises when the HIP library is missing).

Pinning status (see DESIGN.md "Oracle"):

* ``AssociationHead.forward``, ``get_mask_centroid``, ``compute_closest_point``
  are pinned by golden vectors generated from the reference's own Python
  (``tests/golden/make_golden.py`` imports them from /root/reference in the build
  container; the vectors are committed under ``tests/golden/``).
* the PIL resize is pinned by Pillow itself (the reference calls Pillow).
* the Hungarian step is scipy's ``linear_sum_assignment`` (same third party).
* the CSV schema is pinned by ``data/static_dcnn_data.csv`` /
  ``data/dynamic_dcnn_data.csv`` header lines (committed as fixtures).
* everything whose arithmetic lives in detectron2 0.1.2 / torchvision 0.6 /
  OpenCV 4.2 (not installed, source not under /root/reference) is restated from
  the published algorithms: **parity unpinned** for those rows (backbone, FPN,
  RPN, ROI heads, paste, roi_pool, undistort/Lab-gamma), and likewise for the
  SURVEY 8(f) rank-4 additions: ``roi_features.py`` / ``ops.roi_align_legacy``
  (torchvision roi_align aligned=False, F.interpolate) and ``mots.py`` (the MOTS
  writers; pycocotools' RLE format is pinned by the example line the reference
  quotes at dcnn/utils/mots_evaluation.py:13).
"""
