"""GPU: every C-ABI kernel against a plain PyTorch fp64 reference of the same op (kernel_checks.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(fn):
    rows = fn()
    torch.cuda.synchronize()
    bad = [(n, e, t) for n, e, t in rows if not e <= t]
    assert not bad, "\n".join("%s: err %.3e > tol %.1e" % r for r in bad)


def _mk(name):
    def test():
        import kernel_checks as kc
        _run(getattr(kc, name))
    test.__name__ = "test_" + name[6:]
    return test


for _n in ["check_conv_fwd", "check_conv_bwd_data", "check_conv_wgrad", "check_conv_dropout",
           "check_conv_bn_epilogues", "check_conv_rp", "check_dw", "check_zpath", "check_se", "check_na", "check_gattn", "check_ln",
           "check_bn_tail", "check_resample", "check_layout_utils", "check_conv_large", "check_conv_bf16", "check_bf16_storage_rows", "check_bn_shifted_stats", "check_ln_linear", "check_conv_up2", "check_fused_sources_modes", "check_conv_dma3", "check_conv_dma1", "check_conv_dmaM"]:
    globals()["test_" + _n[6:]] = _mk(_n)
