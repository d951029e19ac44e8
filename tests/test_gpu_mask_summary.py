"""-m gpu: the summary level above the validity mask (include/kssd_gpu.h: kssd_gpu_mask_summarise_device,
kssd_gpu_sketch_set_mask_summary).  With it the scan reads one 64-bit word per chunk and fetches the mask words of the lanes that
hold a run-breaking position only; the sketches must not know the difference (iseq2comem.c:213-243: the run counter's resets are
the same bits either way)."""
import numpy as np
import pytest

import public_kssd_amd as K
import test_gpu_sketch as S

pytestmark = pytest.mark.gpu


def test_summary_words_are_the_masks_all_ones_lanes(gpu_ctx):
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    for n_chunks in (1, 3, 4, 5, 257, 4099):
        m = np.full(n_chunks * K.CHUNK_MASKW, 0xFFFFFFFF, dtype=np.uint32)
        # holes: single bits, whole words, whole lanes, a chunk of nothing
        for w in rng.integers(0, len(m), max(1, len(m) // 50)):
            m[w] &= ~np.uint32(1 << int(rng.integers(0, 32)))
        m[rng.integers(0, len(m), max(1, len(m) // 200))] = 0
        if n_chunks > 4:
            m[2 * K.CHUNK_MASKW:3 * K.CHUNK_MASKW] = 0
        d_m = torch.from_numpy(m.view(np.int32)).to(dev)
        d_s = torch.zeros(n_chunks, dtype=torch.int64, device=dev)
        gpu_ctx.mask_summarise_device(d_m, n_chunks, d_s)
        torch.cuda.synchronize()
        got = d_s.cpu().numpy().view(np.uint64)
        lanes = (m.reshape(-1, 2) == 0xFFFFFFFF).all(axis=1).reshape(n_chunks, 64)
        want = (lanes.astype(np.uint64) << np.arange(64, dtype=np.uint64)).sum(axis=1, dtype=np.uint64)
        assert np.array_equal(got, want), n_chunks


def test_device_level_sketch_with_and_without_the_summary(shuf_l3k10):
    """the bench's own batch layout (N at 1e-4, genomes that end inside a chunk), sketched through the plan / phase calls: with the
    summary words the CSR is bit for bit the one without, and scanning with a summary is a per-plan choice (the next plan streams the
    mask again)"""
    import sys, os, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from benchlib.workloads import make_batch
    dev = torch.device("cuda", 0)
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        for G, L in ((24, 300_001), (3, 1_234_567), (40, 20_000)):
            packed, mask, chunk_off, _ = make_batch(G, L, 4, 77 + G, dev)
            n_chunks = int(chunk_off[-1])
            summ = torch.zeros(n_chunks, dtype=torch.int64, device=dev)
            ctx.mask_summarise_device(mask, n_chunks, summ)
            cap = int(G * L / 4096 * 1.5) + 4096
            res = []
            for use in (None, summ, None, summ):
                off = torch.zeros(G + 1, dtype=torch.int64, device=dev)
                ids = torch.zeros(cap, dtype=torch.int32, device=dev)
                for attempt in range(8):
                    ctx.sketch_device(packed, mask, chunk_off, off, ids, cap, d_summary=use)
                    rc, total, bad = ctx.sketch_status()
                    if rc == 0:
                        break
                    assert rc == K.capi.ERR_OVERFLOW, rc
                else:
                    raise AssertionError("sketch kept overflowing")
                res.append((off.cpu().numpy().copy(), ids.cpu().numpy()[:int(total)].copy()))
            assert len(res[0][1]) > G * L / 4096 * 0.8
            for r in res[1:]:
                assert np.array_equal(r[0], res[0][0]) and np.array_equal(r[1], res[0][1])
    finally:
        ctx.close()


def test_read_set_with_and_without_the_summary(shuf_l3k10):
    """a read set is ONE genome of 150-base runs: no lane's neighbourhood is all bases, so the exact-evaluation kernel
    (sketch_exact_kernel: batches with a large genome) settles every candidate's validity itself -- by the summary words where they
    answer, by the mask where they do not; -n 1 and -n 2, ids bit for bit the ones without summary words (which
    tests/test_gpu_configs.py holds against the oracle)"""
    import sys, os, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from benchlib.workloads import make_batch, make_reads_batch
    dev = torch.device("cuda", 0)
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        _, _, _, kept = make_batch(6, 400_000, 3, 5, dev, keep_codes=6, keep_on_device=True)
        rp, rm, rco, _ = make_reads_batch([c for c, _ in kept], 300_000, 99, dev)
        n_chunks = int(rco[-1])
        summ = torch.zeros(n_chunks, dtype=torch.int64, device=dev)
        ctx.mask_summarise_device(rm, n_chunks, summ)
        torch.cuda.synchronize()
        lanes_all_valid = int(np.unpackbits(summ.cpu().numpy().view(np.uint8)).sum())
        assert 0.4 < lanes_all_valid / (n_chunks * 64) < 0.7        # (a run of 150 bases, then a break: 64 / 151 of the lanes hold one)
        cap = int(300_000 * 151 / 4096 * 1.5) + 4096
        for min_occ in (1, 2):
            res = []
            for use in (None, summ):
                off = torch.zeros(2, dtype=torch.int64, device=dev)
                ids = torch.zeros(cap, dtype=torch.int32, device=dev)
                for attempt in range(8):
                    ctx.sketch_device(rp, rm, rco, off, ids, cap, K.SKETCH_KEEP_ZERO | K.SKETCH_NO_CAPACITY, min_occ, d_summary=use)
                    rc, total, bad = ctx.sketch_status()
                    if rc == 0:
                        break
                    assert rc == K.capi.ERR_OVERFLOW, rc
                else:
                    raise AssertionError("sketch kept overflowing")
                res.append(ids.cpu().numpy()[:int(total)].copy())
            assert len(res[0]) > 300 and np.array_equal(res[0], res[1]), (min_occ, len(res[0]), len(res[1]))
    finally:
        ctx.close()


@pytest.fixture
def summarising_ctx(shuf_l3k10, monkeypatch):
    """a context whose host-level calls summarise their resident batch themselves (KSSD_MASK_SUMMARY=1 at creation): every sketch
    call of the tests below scans with the summary words"""
    monkeypatch.setenv("KSSD_MASK_SUMMARY", "1")
    ctx = K.GpuCtx(shuf_l3k10, 0)
    yield ctx
    ctx.close()


def test_the_oracle_suite_of_the_sketch_path_with_summaries(summarising_ctx, shuf_l3k10, monkeypatch):
    """tests of tests/test_gpu_sketch.py and tests/test_gpu_tokenise.py once more, scanning with summary words: edge-case texts (N
    runs, IUPAC, CRLF, records shorter than a k-mer, nine records in one file), clade genomes, FASTQ -n, tiny genomes that share
    chunks' waves, chunks denser than the candidate buffer -- all against the oracle"""
    c = summarising_ctx
    S.test_clade_genomes_l3k10(c, shuf_l3k10)
    S.test_edge_cases_l3k10(c, shuf_l3k10)
    S.test_uniq_mode(c, shuf_l3k10)
    S.test_fastq_min_occ(c, shuf_l3k10)
    S.test_many_tiny_genomes_share_waves_and_chunks(c, shuf_l3k10)
    S.test_chunks_with_more_stage_one_candidates_than_the_buffer_holds(c, shuf_l3k10)
    S.test_empty_batch_and_empty_genome(c, shuf_l3k10)


@pytest.mark.parametrize("params", [(10, 6, 3), (11, 6, 3), (10, 7, 5), (8, 4, 1)])
def test_lengths_around_chunk_and_block_borders_with_summaries(params, monkeypatch):
    monkeypatch.setenv("KSSD_MASK_SUMMARY", "1")   # (the test makes its own contexts)
    S.test_lengths_around_chunk_and_block_borders(params)


def test_large_genome_paths_with_summaries(shuf_l3k10, monkeypatch):
    monkeypatch.setenv("KSSD_MASK_SUMMARY", "1")
    S.test_genomes_sorted_in_lds_in_parts(shuf_l3k10)
    S.test_a_wave_that_owns_more_than_2048_chunks(shuf_l3k10)
    S.test_large_genomes_sorted_by_ranges_of_their_keys()


def test_the_tokeniser_writes_summary_words_with_the_mask(shuf_l3k10):
    """kssd_gpu_tokenise_fasta_device_summary: every set bit is a lane whose 64 positions are all bases (never a wrong one, whatever
    the text: headers, N runs, tiny files, files of hundreds of 16 KiB groups), nearly every such lane is found (the runs of 64
    positions two groups share stay clear), and the batch sketched with these words gives the sketches it gives without"""
    import torch
    import test_gpu_tokenise as T
    dev = torch.device("cuda", 0)
    texts = T._cases()
    ctx = K.GpuCtx(shuf_l3k10, 0)
    try:
        buf, offs, lens = ctx._text_layout(texts)
        co = np.concatenate([[0], np.cumsum([(len(t) + 4095) // 4096 for t in texts])]).astype(np.uint64)
        n_chunks = int(co[-1])
        d_text = torch.from_numpy(buf).to(dev)
        d_packed = torch.zeros(n_chunks * K.CHUNK_WORDS + K.SLACK_WORDS, dtype=torch.int32, device=dev)
        d_mask = torch.zeros(n_chunks * K.CHUNK_MASKW + K.SLACK_WORDS, dtype=torch.int32, device=dev)
        d_summ = torch.full((n_chunks,), -1, dtype=torch.int64, device=dev)           # (zeroed by the call)
        rc, bad, npos = ctx.tokenise_fasta_device(d_text, offs, lens, d_packed, d_mask, co, d_summary=d_summ)
        assert rc == 0 and bad == -1
        exact = torch.zeros(n_chunks, dtype=torch.int64, device=dev)
        ctx.mask_summarise_device(d_mask, n_chunks, exact)
        torch.cuda.synchronize()
        got, want = d_summ.cpu().numpy().view(np.uint64), exact.cpu().numpy().view(np.uint64)
        assert not np.any(got & ~want), "a summary bit is set where the mask holds a run-breaking position"
        n_got, n_want = int(np.unpackbits(got.view(np.uint8)).sum()), int(np.unpackbits(want.view(np.uint8)).sum())
        assert n_want > 50_000 and n_got >= 0.98 * n_want, (n_got, n_want)
        # the same mask without the words: the tokeniser's plain entry point
        m2 = torch.zeros_like(d_mask)
        p2 = torch.zeros_like(d_packed)
        rc, bad, _ = ctx.tokenise_fasta_device(d_text, offs, lens, p2, m2, co)
        assert rc == 0 and torch.equal(m2, d_mask) and torch.equal(p2, d_packed)
        # sketched with the tokeniser's words = sketched without
        cap = int(sum(len(t) for t in texts) / 4096 * 2) + 4096
        res = []
        for use in (None, d_summ):
            off = torch.zeros(len(texts) + 1, dtype=torch.int64, device=dev)
            ids = torch.zeros(cap, dtype=torch.int32, device=dev)
            for attempt in range(8):
                ctx.sketch_device(d_packed, d_mask, co, off, ids, cap, d_summary=use)
                rc, total, _ = ctx.sketch_status()
                if rc == 0:
                    break
                assert rc == K.capi.ERR_OVERFLOW, rc
            res.append((off.cpu().numpy().copy(), ids.cpu().numpy()[:int(total)].copy()))
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and len(res[0][1]) > 500
    finally:
        ctx.close()
