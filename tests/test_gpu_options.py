"""The library's non-default options, seen by the driver's `-m gpu` run: the scheme-level parity tests of test_gpu_codecs.py and the
sign-phase tests of test_gpu_dispatch_parity.py re-run under five option sets through dil_set_option, in process -- no environment, no
rebuild.  (Rounds 2-4 covered these paths only from a builder-side script, scripts/gpu_option_matrix.sh -> profiles/r0*_option_matrix.txt.)
Every test body is the one the default run executes: KAT byte equality (keygen / sign / verify at levels 2 / 3 / 5), the host harness
on random keys and messages, the hardest items of a dispatch-size batch, phase 1 / phase 2 against the oracle."""
import pytest

from tests import test_gpu_codecs as tc
from tests import test_gpu_dispatch_parity as dp

pytestmark = pytest.mark.gpu

OPTION_SETS = {
    "sign_skip=0": {"sign_skip": 0},
    "packed_y=0": {"packed_y": 0},
    "w0w1_plane=0": {"w0w1_plane": 0},
    "fuse_challenge=0": {"fuse_challenge": 0},
    "a24=0,fuse_keygen=0": {"a24": 0, "fuse_keygen": 0},
    "fuse_sib=0": {"fuse_sib": 0},                       # round 6: SampleInBall as a launch of its own in front of the wire-format verify kernel
    "fuse_sib=3": {"fuse_sib": 3},                       # ... and inside that kernel in dil_verify_sig_dev too
    "sign_wake=2": {"sign_wake": 2},                     # ... and with the post "lost": the wait sees the drained stream and falls back to a blocking copy (no hang)
    "sign_wake=0": {"sign_wake": 0},                     # round 6: a signing round's count by copy + event instead of the collect kernel's post into mapped words
    "coop_max=0": {"coop_max": 0},                       # round 5: the lane-per-sponge / two-lane Keccak forms everywhere
    "coop_max=2^30,sign_early=0": {"coop_max": 1 << 30, "sign_early": 0},
}


@pytest.fixture(params=list(OPTION_SETS))
def options(request, gpu):
    from dilithium_amd import api
    want = OPTION_SETS[request.param]
    saved = {k: api.get_option(k) for k in want}
    for k, v in want.items():
        api.set_option(k, v)
    yield request.param
    for k, v in saved.items():
        api.set_option(k, v)


@pytest.mark.parametrize("level", [2, 3, 5])
def test_kat_keygen_sign_verify(gpu, options, kat_msgs, level):
    tc.test_keygen_kat(gpu, level)
    tc.test_sign_wire_kat(gpu, level, kat_msgs)
    tc.test_verify_sig_wire_kat(gpu, level, False, kat_msgs)
    tc.test_verify_sig_wire_kat(gpu, level, True, kat_msgs)


@pytest.mark.parametrize("level", [2, 3, 5])
def test_small_batches_and_shared_key(gpu, options, kat_msgs, level):
    tc.test_sign_single_and_small_batches(gpu, level, kat_msgs)
    tc.test_sign_shared_key_many_messages(gpu, level, kat_msgs)


@pytest.mark.parametrize("level", [2, 3, 5])
def test_random_keys_vs_host_harness(gpu, options, level):
    tc.test_random_keys_and_messages_vs_host_harness(gpu, level)


@pytest.mark.parametrize("level", [3])
def test_dispatch_size_batch_hardest_items(gpu, options, level):
    tc.test_hardest_items_of_a_dispatch_size_batch_vs_host_harness(gpu, level)


@pytest.mark.parametrize("level", [2, 3, 5])
def test_sign_phases_vs_oracle(gpu, options, oracle, level):
    dp.test_sign_phases_at_dispatch_size_vs_oracle(gpu, oracle, level, True)
