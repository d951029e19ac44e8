"""device tokeniser timing: FASTQ (one 1.9 GB read set) and FASTA (256 genomes of 5 Mb) texts resident in HBM.
usage (GPU box): rocprofv3 --kernel-trace --stats ... -- python3 profiles/tok_probe.py"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch, numpy as np
import public_kssd_amd as K
from synth import fastq_records, fasta_text
dev = torch.device("cuda", 0)
shuf = K.Shuf.generate(10, 6, 3, seed=20260101)
ctx = K.GpuCtx(shuf, 0)
rng = np.random.default_rng(1)
fq = fastq_records(rng.integers(0, 4, (6_000_000, 150), dtype=np.uint8))
fa = [fasta_text(rng.integers(0, 4, 5_000_000, dtype=np.uint8)) for _ in range(4)] * 64
for name, texts, isq in (("fastq", [fq], True), ("fasta", fa, False)):
    buf, offs, lens = ctx._text_layout(texts)
    co = np.concatenate([[0], np.cumsum([(len(t) + 4095) // 4096 for t in texts])]).astype(np.uint64)
    d_text = torch.from_numpy(buf).to(dev)
    n = int(co[-1])
    tp = torch.zeros(n * K.CHUNK_WORDS + K.SLACK_WORDS, dtype=torch.int32, device=dev)
    tm = torch.zeros(n * K.CHUNK_MASKW + K.SLACK_WORDS, dtype=torch.int32, device=dev)
    best = 1e9
    for it in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = ctx.tokenise_fasta_device(d_text, offs, lens, tp, tm, co, fastq=isq)
        best = min(best, time.perf_counter() - t0)
        assert r[0] == 0
    print("%s: %d files, %.2f GB of text: %.2f ms = %.0f GB/s" % (name, len(texts), len(buf) / 1e9, best * 1e3, len(buf) / 1e9 / best))
