"""GPU parity of on-device message hashing: mu = SHAKE256(tr || M, 64) for ragged messages (what the reference's top level
absorbs itself -- rtl_src/expandmask_ext.v:131-185, bus order rtl_tb/tb_sign_top.v:57-69, tb_verify_top.v:58-68), and the
KATs driven from their ACTUAL inputs: (sk, M) -> signature bytes, (pk, M, sig) -> accept."""
import hashlib

import numpy as np
import pytest

from tests.test_gpu_codecs import cu, kat_wire

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("level", [2, 3, 5])
def test_mu_of_all_kat_messages(gpu, level, kat_msgs):
    """the 100 KAT messages (33 ... 3300 bytes, one ragged batch, a tr per item) -> mu == hashlib"""
    from dilithium_amd import api
    k, *_ = kat_wire(level)
    blob, offs, lens = api.pack_messages(kat_msgs)
    mu = api.mu(cu(gpu, k["tr"]), blob, offs, lens).cpu().numpy()
    for i, m in enumerate(kat_msgs):
        assert mu[i].tobytes() == hashlib.shake_256(k["tr"][i].tobytes() + m).digest(64), (i, len(m))


def test_mu_block_boundaries_and_alignment(gpu):
    """message lengths around every padding case of the 136-byte rate (tr fills the first 32 bytes): empty, one byte,
    word and block boundaries +-1, multi-block; packed back to back, so most messages start at odd addresses; one tr"""
    from dilithium_amd import api
    rng = np.random.default_rng(1)
    lens = [0, 1, 7, 8, 9, 95, 96, 97, 103, 104, 105, 111, 112, 135, 136, 137, 239, 240, 241, 272, 1000, 3299, 3300, 5000] + \
        list(range(200, 264))
    msgs = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in lens]
    tr = rng.integers(0, 256, (1, 32), dtype=np.uint8)
    blob, offs, ln = api.pack_messages(msgs)
    mu = api.mu(cu(gpu, tr), blob, offs, ln).cpu().numpy()
    for i, m in enumerate(msgs):
        assert mu[i].tobytes() == hashlib.shake_256(tr[0].tobytes() + m).digest(64), len(m)


@pytest.mark.parametrize("level", [2, 3, 5])
def test_kat_sign_and_verify_from_messages(gpu, level, kat_msgs):
    """(sk, M) -> the KAT signature bytes with the KAT attempt counts; (pk, M, sig) -> accept; another message -> reject"""
    from dilithium_amd import api
    k, pk, sk, sig = kat_wire(level)
    blob, offs, lens = api.pack_messages(kat_msgs)
    got, att = api.sign_msg(cu(gpu, sk), blob, offs, lens, level)
    assert (att.cpu().numpy() == k["attempts"]).all()
    assert (got.cpu().numpy() == sig).all()
    assert (api.verify_msg(cu(gpu, pk), got, blob, offs, lens, level).cpu().numpy() == 0).all()
    other = list(kat_msgs)
    other[5] = other[5][:-1] + bytes([other[5][-1] ^ 1])
    other[9] = other[9] + b"!"
    b2, o2, l2 = api.pack_messages(other)
    v = api.verify_msg(cu(gpu, pk), got, b2, o2, l2, level).cpu().numpy()
    assert set(np.nonzero(v)[0]) == {5, 9}


@pytest.mark.parametrize("level", [3])
def test_shared_key_messages(gpu, level, kat_msgs):
    """one signer, 3000 ragged messages: sign_msg (tr read once from the key) == sign on host-hashed mu; verify_msg with
    one pk (tr = SHAKE256(pk) on the device) accepts all"""
    from dilithium_amd import api
    k, pk, sk, _ = kat_wire(level)
    rng = np.random.default_rng(4)
    msgs = [rng.integers(0, 256, int(rng.integers(0, 400)), dtype=np.uint8).tobytes() for _ in range(3000)]
    msgs[0] = kat_msgs[0]
    blob, offs, lens = api.pack_messages(msgs)
    sig, _ = api.sign_msg(cu(gpu, sk[:1]), blob, offs, lens, level, shared_sk=True)
    mu = np.stack([np.frombuffer(hashlib.shake_256(k["tr"][0].tobytes() + m).digest(64), dtype=np.uint8) for m in msgs])
    ref, _ = api.sign(cu(gpu, sk[:1]), cu(gpu, mu), level, shared_sk=True)
    assert (sig == ref).all()
    assert sig[0].cpu().numpy().tobytes() == k["ctilde"][0].tobytes() + k["z"][0].tobytes() + k["h"][0].tobytes()
    assert int(api.verify_msg(cu(gpu, pk[:1]), sig, blob, offs, lens, level, shared_pk=True).abs().sum()) == 0
