"""-m gpu: the collective path of a frame-sharded run on the RCCL backend ("nccl" on ROCm), as far as one GPU can take it:
a one-rank process group on cuda:0 running exactly the calls of bench.py / tools/run_sequence.py -- the record gather
(``sharding.gather_records``: a 16-byte ``all_gather_into_tensor`` of sizes + one ``dist.gather`` of the compact payloads, on device tensors), the barrier and the max-over-ranks reduction of
the elapsed time (float64).  More ranks need more GPUs (RCCL refuses two ranks on one device); the multi-rank logic itself is
covered on gloo (tests/test_sharding_gloo.py).  Reference loop being sharded: /root/reference/dcnn/scripts/tests/visualize_uav.py:186-221."""
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_record_gather_and_timing_reduction_on_rccl():
    import torch.distributed as dist
    from apse_uav_amd.sharding import gather_records, record_len
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        rng = np.random.default_rng(0)
        recs = []
        for n in (3, 0, 8):
            recs.append(dict(boxes=rng.random((n, 4), dtype=np.float32) * 100, scores=rng.random(n, dtype=np.float32),
                             classes=rng.integers(0, 4, n).astype(np.int64), centroids=rng.integers(0, 2000, (n, 2)).astype(np.int32),
                             mass=rng.integers(1, 5000, n).astype(np.int32), rects=rng.integers(0, 2000, (n, 4)).astype(np.int32),
                             closest=rng.integers(0, 2000, (n, n, 2)).astype(np.int32), embeddings=rng.random((n, 128), dtype=np.float32)))
        dist.barrier()
        packed, nrec = gather_records(recs, 0, 1, dev, unpack=False)
        assert nrec == 3 and packed.shape == (record_len(3) + record_len(0) + record_len(8),)
        got = gather_records(recs, 0, 1, dev)
        assert len(got) == 3
        for a, b in zip(recs, got):
            assert len(b["scores"]) == len(a["scores"])
            assert np.array_equal(b["boxes"], a["boxes"]) and np.array_equal(b["embeddings"], a["embeddings"])
            assert np.array_equal(np.asarray(b["closest"]), a["closest"])
        tt = torch.tensor([1.25], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        assert float(tt) == 1.25
        dist.barrier()
    finally:
        dist.destroy_process_group()
