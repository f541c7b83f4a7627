"""The C restatement of the codebook search (oracle/vq_argmin.c): CPU test against the torch oracle and
the golden fixture; GPU test: the HIP kernel's indices AND distances against it bit for bit."""
import ctypes
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    import importlib.util
    spec = importlib.util.spec_from_file_location("oracle_build", os.path.join(ROOT, "oracle", "build_oracle.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    lib = ctypes.CDLL(m.build())
    lib.vq_argmin_oracle.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p]
    return lib


def c_argmin(z, E):
    lib = _lib()
    z = np.ascontiguousarray(z, np.float32)
    E = np.ascontiguousarray(E, np.float32)
    N = z.shape[0]
    idx, bd, mg = np.empty(N, np.int32), np.empty(N, np.float32), np.empty(N, np.float32)
    lib.vq_argmin_oracle(z.ctypes.data, N, E.ctypes.data, idx.ctypes.data, bd.ctypes.data, mg.ctypes.data)
    return idx, bd, mg


def test_c_oracle_matches_torch_oracle_and_golden(golden_dir, vq_sd):
    from oracle import ref_cpu
    E = vq_sd["listener_vq.quantize.embedding.weight"]
    g = np.load(os.path.join(golden_dir, "vq_encode_T300.npz"))
    idx, bd, mg = c_argmin(g["z"], E.numpy())
    assert np.array_equal(idx, g["idx"].astype(np.int32))
    ref_idx, d = ref_cpu.vq_quantize(torch.from_numpy(g["z"]), E)
    assert np.abs(bd - d.min(1).values.numpy()).max() < 1e-3
    # ties resolve to the first index
    z = np.zeros((1, 128), np.float32)
    E2 = np.zeros((512, 128), np.float32)
    assert c_argmin(z, E2)[0][0] == 0


@pytest.mark.gpu
def test_hip_argmin_is_bit_exact_with_c_oracle(vq_sd):
    from dimx import engine, lib
    gen = torch.Generator().manual_seed(3)
    z = torch.randn(777, 128, generator=gen) * 0.7
    E = vq_sd["listener_vq.quantize.embedding.weight"]
    e = engine.Engine("cuda:0", lib.MODE_PARITY_F32)
    e.load_state_dict({k: v for k, v in vq_sd.items() if k.startswith("listener_vq.")})
    idx, bd, mg = e.vq_argmin(1, z.cuda(), with_stats=True)
    cidx, cbd, cmg = c_argmin(z.numpy(), E.numpy())
    assert np.array_equal(idx.cpu().numpy(), cidx)
    assert np.array_equal(bd.cpu().numpy().view(np.uint32), cbd.view(np.uint32)), "distances differ in the last bit"
    assert np.array_equal(mg.cpu().numpy().view(np.uint32), cmg.view(np.uint32))
