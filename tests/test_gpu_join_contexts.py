"""-m gpu: the records of dmnd_extend stay in HBM for the block join (round 6: dmnd_extend_records_device, dmnd_join_contexts_device).
Two reference blocks on two contexts, the same queries: the join over the device-resident records must equal the host join of the
records dmnd_extend returned (dmnd_join_blocks, pinned on the reference's heap merge in tests/test_join_blocks.py), and the device
copy must be the returned records byte for byte (the host's e-values and bit scores written back)."""
import ctypes

import numpy as np
import pytest
import torch

from diamond_amd import hip, synth, workload

pytestmark = pytest.mark.gpu


def test_device_resident_records_join_equals_host_join():
    assert torch.cuda.is_available()
    db, doff, q, qoff = synth.generate(300, members=10, queries=400, seed=5)
    qd, ql = workload.sequence_set(q, qoff)
    half = (len(doff) - 1) // 2
    params = hip.default_params()
    params.db_letters = float(doff[-1])
    ctxs, recs, offs = [], [], []
    try:
        for a, b in ((0, half), (half, len(doff) - 1)):
            td, tl = workload.sequence_set(db[doff[a]:doff[b]], doff[a:b + 1] - doff[a])
            c = hip.Context(params=params)
            c.upload_block(hip.QUERY, qd, ql)
            c.upload_block(hip.TARGET, td, tl)
            hits = c.seed_search(hip.seed_params_fast(threads=4))
            m = c.extend(qd, td, hits, threads=4)[0]
            ptr, n = c.extend_records_device()
            assert n == len(m) > 0 and ptr, "the call's records are not complete in HBM"
            torch.cuda.synchronize()
            got = np.empty(n, dtype=hip.MATCH_DTYPE)
            rt = ctypes.CDLL("libamdhip64.so")
            rt.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
            assert rt.hipMemcpy(got.ctypes.data, ctypes.c_void_p(ptr), n * hip.MATCH_DTYPE.itemsize, 2) == 0      # hipMemcpyDeviceToHost
            assert got.tobytes() == np.ascontiguousarray(m).tobytes()
            ctxs.append(c); recs.append(m.copy()); offs.append(a)
        jc = hip.Context(params=params)
        ctxs.append(jc)
        joined = jc.join_contexts_device(ctxs[:2], offs, 25, max_query=399)
        host = np.concatenate(recs)
        at = 0
        for m, a in zip(recs, offs):
            host["target"][at:at + len(m)] += np.uint32(a)
            at += len(m)
        want = hip.join_blocks(host, 25)
        assert len(joined) == len(want) > 0
        assert joined.tobytes() == want.tobytes()
    finally:
        for c in ctxs:
            c.close()
