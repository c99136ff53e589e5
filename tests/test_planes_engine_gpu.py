"""The fp32 engine on operand planes (csrc/gemm_xp.hip + the plane-writing LayerNorm / attention / GEMM epilogues) against the
same engine on the register-split kernels (SIMX_F32_PLANES=0): one process per mode (the switch is read once per process),
same weights, ragged batch, dropout on and off, kept activations and gradient checkpointing, full and [CLS]-only last layer.
Both are f32-grade arithmetic (hi.lo + lo.hi + hi.hi of the same splits, different summation order): embeddings within 2e-5,
gradients within 1e-4 of their scale.  The reference goldens (tests/test_encoder_gpu.py, hot fixture) pin the plane path itself."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp, tag, planes, ckpt, cls_only, dropout):
    out = os.path.join(tmp, "%s.npz" % tag)
    env = dict(os.environ, SIMX_F32_PLANES="1" if planes else "0", SIMX_F32_PLANES_MIN_TILES="1")
    subprocess.run([sys.executable, os.path.join(HERE, "_planes_ab_worker.py"), out, str(ckpt), str(cls_only), str(dropout)], check=True,
                   env=env, timeout=600)
    return np.load(out)


@pytest.mark.parametrize("ckpt,cls_only,dropout", [(0, 0, 0.0), (0, 1, 0.1), (1, 1, 0.1), (1, 0, 0.0)])
def test_planes_engine_equals_split_engine(dev, tmp_path, ckpt, cls_only, dropout):
    a = _run(str(tmp_path), "planes", True, ckpt, cls_only, dropout)
    b = _run(str(tmp_path), "split", False, ckpt, cls_only, dropout)
    assert np.isfinite(a["emb"]).all() and np.isfinite(a["grad"]).all()
    e = np.abs(a["emb"] - b["emb"]).max()
    assert e <= 2e-5 * max(1.0, np.abs(b["emb"]).max()), e
    g = np.abs(a["grad"] - b["grad"]).max()
    assert g <= 1e-4 * np.abs(b["grad"]).max(), (g, np.abs(b["grad"]).max())
    assert np.abs(a["grad"]).max() > 0
