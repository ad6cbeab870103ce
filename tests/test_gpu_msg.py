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


def test_message_reference_outside_the_blob_is_flagged_not_read(gpu, kat_msgs):
    """an item whose (offset, length) leaves the blob is never read past it: mu flags it (and hashes an empty message),
    sign_msg voids it (attempts -1, zero signature), verify_msg rejects it with bit 3; the other items are untouched.  An
    all-empty batch may pass (NULL, 0) as its blob."""
    from dilithium_amd import api
    level = 3
    k, pk, sk, _ = kat_wire(level)
    msgs = [bytes([i]) * (10 + i) for i in range(8)]
    blob, offs, lens = api.pack_messages(msgs)
    good_sig, good_att = api.sign_msg(cu(gpu, sk[:1]), blob, offs, lens, level, shared_sk=True)
    o2, l2 = offs.clone(), lens.clone()
    o2[2] = blob.numel() - 3          # runs 3 + ... past the end
    l2[5] = 1 << 30                   # absurd length
    o2[6] = (1 << 62)                 # absurd offset (offset + length would wrap)
    bad = gpu.full((8,), -7, dtype=gpu.int32, device="cuda")
    mu = api.mu(cu(gpu, k["tr"][:1]), blob, o2, l2, bad=bad).cpu().numpy()
    assert bad.cpu().numpy().tolist() == [0, 0, 1, 0, 0, 1, 1, 0]
    empty = hashlib.shake_256(k["tr"][0].tobytes()).digest(64)
    for i in range(8):
        want = empty if i in (2, 5, 6) else hashlib.shake_256(k["tr"][0].tobytes() + msgs[i]).digest(64)
        assert mu[i].tobytes() == want, i
    sig, att = api.sign_msg(cu(gpu, sk[:1]), blob, o2, l2, level, shared_sk=True)
    att = att.cpu().numpy()
    for i in range(8):
        if i in (2, 5, 6):
            assert att[i] == -1 and int(sig[i].sum()) == 0
        else:
            assert att[i] == int(good_att[i]) and gpu.equal(sig[i], good_sig[i])
    v = api.verify_msg(cu(gpu, pk[:1]), good_sig, blob, o2, l2, level, shared_pk=True).cpu().numpy()
    assert [int(x) & 8 for x in v] == [0, 0, 8, 0, 0, 8, 8, 0] and all(v[i] == 0 for i in (0, 1, 3, 4, 7))
    # every message empty: no blob at all
    eb, eo, el = api.pack_messages([b""] * 5)
    assert eb.numel() == 0
    s5, a5 = api.sign_msg(cu(gpu, sk[:1]), eb, eo, el, level, shared_sk=True)
    mu0 = np.frombuffer(empty, dtype=np.uint8)[None].repeat(5, 0)
    ref, _ = api.sign(cu(gpu, sk[:1]), cu(gpu, mu0), level, shared_sk=True)
    assert gpu.equal(s5, ref) and (a5 > 0).all()
    assert int(api.verify_msg(cu(gpu, pk[:1]), s5, eb, eo, el, level, shared_pk=True).abs().sum()) == 0


def test_verify_msg_many_keys_uses_the_helper_stream_once(gpu):
    """dil_verify_msg_dev with more keys than the few-keys path takes (> 32768 A-polynomials): the call hands ITS helper-stream
    fork down to the verification instead of try-locking the helper's mutex a second time on the same thread; verdicts equal
    those of verify_sig on host-hashed mu, tampered items rejected"""
    from dilithium_amd import api
    level, n = 3, 1400                       # 1400 x 30 = 42000 polynomials of A
    rng = np.random.default_rng(31)
    seed = cu(gpu, rng.integers(0, 256, (n, 32), dtype=np.uint8))
    pk, sk = api.keygen(seed, level)
    msgs = [rng.integers(0, 256, int(rng.integers(0, 120)), dtype=np.uint8).tobytes() for _ in range(n)]
    blob, offs, lens = api.pack_messages(msgs)
    sig, _ = api.sign_msg(sk, blob, offs, lens, level)
    bad = sig.clone()
    bad[7, 100] ^= 1
    bad[1399, 5] ^= 64
    v = api.verify_msg(pk, bad, blob, offs, lens, level).cpu().numpy()
    assert set(np.nonzero(v)[0]) == {7, 1399}
    tr = api.shake256(pk, 32)
    mu = api.mu(tr, blob, offs, lens)
    assert (api.verify_sig(pk, bad, mu, level).cpu().numpy() == v).all()
