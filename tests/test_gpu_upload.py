"""The reads' way to the device (sp_seqset_upload_format / _async / _wait, sp_bam_last_seq4): the three input formats give the same set,
an upload under way does not disturb the context's other work, one over-long read does not cost the sample."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rnd(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(list(alphabet), n))


def self_alignments(ctx, A, B, n):
    idx = np.arange(n, dtype=np.uint32)
    diag, votes = ctx.anchor_batch(A, B, idx, idx)
    return ctx.align_batch(A, B, idx, idx, diag, 64)


def test_formats_give_the_same_set(pkg, gpu_ctx):
    rng = np.random.default_rng(5)
    seqs = [rnd(rng, int(n)) for n in rng.integers(40, 9000, 300)] + [rnd(rng, 16), rnd(rng, 17), rnd(rng, 31), rnd(rng, 33)]
    ascii_set = gpu_ctx.upload(seqs)
    b4 = gpu_ctx.upload_format(pkg.ffi.SP_SEQ_BAM4, *pkg.ffi.encode_bam4(seqs))
    p2 = gpu_ctx.upload_format(pkg.ffi.SP_SEQ_PACKED2, *pkg.ffi.encode_packed2(seqs))
    n = len(seqs)
    for other in (b4, p2):
        assert other.n == n and other.lengths.tolist() == [len(s) for s in seqs]
        out = self_alignments(gpu_ctx, ascii_set, other, n)
        assert (out["ok"] == 1).all() and (out["nm"] == 0).all()
        assert (out["a_end"] - out["a_start"]).tolist() == [len(s) for s in seqs] and (out["b_end"] - out["b_start"]).tolist() == [len(s) for s in seqs]
    # one base changed in the 4-bit form: exactly one edit
    seqs2 = list(seqs)
    seqs2[7] = seqs2[7][:500] + ("A" if seqs2[7][500] != "A" else "C") + seqs2[7][501:]
    out = self_alignments(gpu_ctx, ascii_set, gpu_ctx.upload_format(pkg.ffi.SP_SEQ_BAM4, *pkg.ffi.encode_bam4(seqs2)), n)
    assert out["nm"].tolist() == [1 if i == 7 else 0 for i in range(n)]


def test_ambiguity_codes_are_n_in_both_forms(pkg, gpu_ctx):
    rng = np.random.default_rng(6)
    base = [rnd(rng, 600) for _ in range(8)]
    with_n = [s[:300] + c + s[301:] for s, c in zip(base, "NRYKMSWB")]
    clean = gpu_ctx.upload(base)
    for other in (gpu_ctx.upload(with_n), gpu_ctx.upload_format(pkg.ffi.SP_SEQ_BAM4, *pkg.ffi.encode_bam4(with_n))):
        out = self_alignments(gpu_ctx, clean, other, 8)
        assert (out["nm"] == 1).all()                        # the ambiguous base matches nothing
        out = self_alignments(gpu_ctx, other, other, 8)
        assert (out["nm"] == 1).all()                        # not even itself (N never matches: DESIGN.md 3.2)


def test_async_upload_runs_beside_the_contexts_work(pkg, gpu_ctx):
    rng = np.random.default_rng(7)
    first = [rnd(rng, 3000) for _ in range(400)]
    A = gpu_ctx.upload(first)
    big = [rnd(rng, 8000) for _ in range(4000)]             # 32 MB of ASCII: four staging chunks
    blob, offs = pkg.ffi._concat(big)
    pending = gpu_ctx.upload_format(pkg.ffi.SP_SEQ_ASCII, blob, offs, wait=False)
    out = self_alignments(gpu_ctx, A, A, len(first))       # the context keeps working while the bytes travel
    assert (out["nm"] == 0).all()
    B = pending.wait()
    assert B.n == len(big)
    pick = rng.choice(len(big), 300, replace=False)
    C_ = gpu_ctx.upload([big[i] for i in pick])
    idx_a, idx_b = np.arange(300, dtype=np.uint32), pick.astype(np.uint32)
    diag, _v = gpu_ctx.anchor_batch(C_, B, idx_a, idx_b)
    out = gpu_ctx.align_batch(C_, B, idx_a, idx_b, diag, 64)
    assert (out["ok"] == 1).all() and (out["nm"] == 0).all() and ((out["b_end"] - out["b_start"]) == 8000).all()
    # a second upload started while one is in flight waits for it; both end complete
    p1 = gpu_ctx.upload_format(pkg.ffi.SP_SEQ_ASCII, blob, offs, wait=False)
    p2 = gpu_ctx.upload_format(pkg.ffi.SP_SEQ_BAM4, *pkg.ffi.encode_bam4(first), wait=False)
    assert p2.wait().n == len(first) and p1.wait().n == len(big)
    out = self_alignments(gpu_ctx, A, p2, len(first))
    assert (out["nm"] == 0).all()


def test_freed_sets_hand_their_buffers_to_the_next_ones(pkg):
    """sp_seqset_free keeps the device buffers in the context for the next upload (hipFree would wait for every stream of the device): sets uploaded into buffers other
    sets left behind -- larger ones, smaller ones, with and without an N plane or a k-mer index -- hold exactly their own reads; a context that is destroyed before one of its
    sets leaves that set its buffers (the set frees them itself)."""
    rng = np.random.default_rng(17)
    ctx = pkg.Context(0)
    previous = None
    for round_ in range(12):
        n = int(rng.integers(3, 120))
        seqs = [rnd(rng, int(m)) for m in rng.integers(20, 6000, n)]
        if round_ % 3 == 0:                                                      # a few Ns: the set gets an N plane (a fourth buffer)
            seqs = [q if len(q) < 40 else "".join("N" if i in (7, len(q) // 2, len(q) - 9) else c for i, c in enumerate(q)) for q in seqs]
        S = ctx.upload(seqs) if round_ % 2 else ctx.upload_format(pkg.ffi.SP_SEQ_BAM4, *pkg.ffi.encode_bam4(seqs))
        T = ctx.upload(seqs)                                                     # a second copy: the pair aligns base for base (N positions count as edits on both sides)
        idx = np.arange(n, dtype=np.uint32)
        diag, votes = ctx.anchor_batch(S, T, idx, idx)                            # (builds S's k-mer index: three more buffers that go back to the cache)
        out = ctx.align_batch(S, T, idx, idx, np.zeros(n, np.int32), 255)
        assert (out["ok"] == 1).all(), round_
        assert (out["a_end"] - out["a_start"]).tolist() == [len(q) for q in seqs], round_
        assert out["nm"].tolist() == [q.count("N") for q in seqs], round_
        if previous is not None:
            previous.close()                                                    # the set of the round before goes back while this round's sets are alive
        previous = S
        T.close()
    last = ctx.upload([rnd(rng, 500)])
    ctx.close()                                                                 # the context first ...
    last.close(); previous.close()                                              # ... then sets it had handed buffers to


def test_an_over_long_read_is_left_out_not_fatal(pkg, gpu_ctx):
    rng = np.random.default_rng(8)
    seqs = [rnd(rng, 2000), rnd(rng, 70000), rnd(rng, 2500), ""]
    S = gpu_ctx.upload(seqs)
    assert S.skipped == 1
    import ctypes as C
    ln = C.c_uint32(7)
    assert pkg.ffi.lib().sp_seqset_length(S._h, 1, C.byref(ln)) == 0 and ln.value == 0
    T = gpu_ctx.upload([seqs[0], seqs[2]])
    idx_a, idx_b = np.array([0, 1], np.uint32), np.array([0, 2], np.uint32)
    diag, _v = gpu_ctx.anchor_batch(T, S, idx_a, idx_b)
    out = gpu_ctx.align_batch(T, S, idx_a, idx_b, diag, 64)
    assert (out["ok"] == 1).all() and (out["nm"] == 0).all()


def test_upload_argument_errors(pkg, gpu_ctx):
    blob, offs, lens = pkg.ffi.encode_bam4(["ACGT" * 10])
    with pytest.raises(pkg.StarphaseError):
        gpu_ctx.upload_format(pkg.ffi.SP_SEQ_BAM4, blob, offs, np.array([1000], np.uint32))      # longer than its bytes
    with pytest.raises(pkg.StarphaseError):
        gpu_ctx.upload_format(7, blob, offs, lens)


def test_group_of_one_rank_gathers_through_rccl(pkg, gpu_ctx):
    """sp_group_* / sp_gather_results on the one GPU of this box: a group of a single rank still goes through ncclCommInitRank and
    ncclAllGather (the N > 1 node is the driver's to run; two ranks on one device are refused by RCCL itself)"""
    from pb_starphase_amd import shard
    uid = pkg.ffi.group_unique_id()
    assert uid.shape == (128,) and uid.any()
    g = pkg.ffi.Group(gpu_ctx, uid, 0, 1)
    import ctypes as C
    r, n = C.c_int32(-1), C.c_int32(-1)
    assert pkg.ffi.lib().sp_group_size(g._h, C.byref(r), C.byref(n)) == 0 and (r.value, n.value) == (0, 1)
    rec = np.zeros(5, shard.CALL_DTYPE)
    for k in range(5):
        rec[k] = (3, k, 10 + k, 20 + k)
    out = g.gather(rec)
    assert out.shape == (1, 5) and out[0].tolist() == rec.tolist()
    table = shard.gather_calls(rec[::-1].copy(), group=g, same_count=True)
    assert table.tolist() == rec.tolist()
    assert shard.gather_calls(rec, group=g).tolist() == rec.tolist()
    big = np.arange(100000, dtype=np.int64)
    assert (g.gather(big)[0] == big).all()
    g.close()
