"""Two ranks of the latent-sharded engine on ONE MI355X (both on cuda:0, collectives staged through gloo): the HIP
kernels' shard arguments (lat_begin / local latent counts), the candidate merge and the partial decode must
reproduce the single-process oracle exactly as the CPU double does in tests/test_tp_gloo.py.  RCCL itself is not
exercised here (it needs one GPU per rank); the collectives' call sites and shapes are."""
import pytest

from test_tp_gloo import CASES, run_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", list(CASES))
def test_tp2_hip_matches_single_process_oracle(case, tmp_path):
    run_case(case, tmp_path, device="cuda")


def test_tp2_hip_checkpoint_resume(tmp_path):
    """Save after two steps, resume in a fresh trainer (HIP Adam state incl. its step counter), third step unchanged."""
    run_case("clt_skip_scale_norm_div", tmp_path, device="cuda", resume=True)
