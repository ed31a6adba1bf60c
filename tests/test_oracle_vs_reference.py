"""CPU, build container only: oracle vs the real reference imported from /root/reference."""
import pytest
import torch

from tools.ref_import import import_reference_lmnet, reference_available
from tools.detweights import det_input, fill_module
from helpers import no_dropout, rel_err

pytestmark = pytest.mark.skipif(not reference_available(), reason="/root/reference not present (GPU box)")


def test_eval_and_train_forward_match_reference():
    from oracle.lmnet_ref import LM_Net
    Ref = import_reference_lmnet()
    r, o = Ref(3, 2), LM_Net(3, 2)
    fill_module(r, seed=3)
    o.load_state_dict(r.state_dict())
    x = det_input((2, 3, 48, 64), "vsref/x")
    r.eval(); o.eval()
    with torch.no_grad():
        assert rel_err(o(x), r(x)) < 1e-5
    no_dropout(r); no_dropout(o)
    r.train(); o.train()
    assert rel_err(o(x), r(x)) < 1e-5
