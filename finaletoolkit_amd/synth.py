"""
Seeded synthetic cfDNA fragments (the workload generator of BASELINE.md
section 4 / SURVEY.md section 8-d).  NumPy version: used by the parity tests
and small benches; bench.py has a device-side generator of the same
distribution for whole-genome sizes.
"""
from __future__ import annotations

import numpy as np

# b37 / hg19 primary contigs (tests/data/b37.chrom.sizes of the reference)
B37_SIZES = {
    "1": 249250621, "2": 243199373, "3": 198022430, "4": 191154276, "5": 180915260, "6": 171115067,
    "7": 159138663, "8": 146364022, "9": 141213431, "10": 135534747, "11": 135006516, "12": 133851895,
    "13": 115169878, "14": 107349540, "15": 102531392, "16": 90354753, "17": 81195210, "18": 78077248,
    "19": 59128983, "20": 63025520, "21": 48129895, "22": 51304566, "X": 155270560, "Y": 59373566,
}
SEED_BASE = 20260723


def n_fragments(contig_len: int, depth: float) -> int:
    """N_frag(contig) = round(depth * contig_len / 300)  (2 x 150 bp pairs)."""
    return int(round(depth * contig_len / 300.0))


def synth_contig(contig_len: int, depth: float = 30.0, seed: int = SEED_BASE, n: int | None = None):
    """Return start-sorted SoA columns (start i32, end i32, mapq u8, strand u8)."""
    rng = np.random.default_rng(seed)
    if n is None:
        n = n_fragments(contig_len, depth)
    start = rng.integers(0, max(contig_len - 1000, 1), size=n, dtype=np.int64)
    u = rng.random(n)
    length = np.where(
        u < 0.85, rng.normal(167.0, 12.0, n),
        np.where(u < 0.97, rng.normal(334.0, 25.0, n), rng.uniform(30.0, 600.0, n)))
    length = np.clip(np.rint(length), 30, 1000).astype(np.int64)
    end = start + length
    mapq = np.where(rng.random(n) < 0.85, 60, rng.integers(0, 60, size=n)).astype(np.uint8)
    strand = (rng.random(n) < 0.5).astype(np.uint8)
    order = np.lexsort((end, start))
    return (start[order].astype(np.int32), end[order].astype(np.int32), mapq[order], strand[order])


def tiling_windows(contig_len: int, width: int):
    """Non-overlapping windows [k*width, min((k+1)*width, contig_len))."""
    ws = np.arange(0, contig_len, width, dtype=np.int64)
    we = np.minimum(ws + width, contig_len)
    return ws.astype(np.int32), we.astype(np.int32)


# ---- whole-genome workload pieces (bench.py, tools/, the full-size GPU tests) ---------------------------------
_WINDOW = 100_000


def gen_contig_device(torch, dev, contig_len, n, seed):
    """Seeded device-side generator of the BASELINE.md mixture; start-sorted SoA tensors."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    start = torch.randint(0, max(contig_len - 1000, 1), (n,), generator=g, device=dev, dtype=torch.int64)
    u = torch.rand(n, generator=g, device=dev)
    z = torch.randn(n, generator=g, device=dev)
    v = torch.rand(n, generator=g, device=dev)
    length = torch.where(u < 0.85, 167.0 + 12.0 * z, torch.where(u < 0.97, 334.0 + 25.0 * z, 30.0 + 570.0 * v))
    length = torch.clamp(torch.round(length), 30, 1000).to(torch.int64)
    del u, z, v
    key, _ = torch.sort(start * 2048 + length)  # sort by (start, end)
    del start, length
    s = (key >> 11).to(torch.int32)
    e = (s + (key & 2047).to(torch.int32)).contiguous()
    del key
    m = torch.rand(n, generator=g, device=dev)
    mapq = torch.where(m < 0.85, torch.full((n,), 60, device=dev, dtype=torch.int64),
                       torch.randint(0, 60, (n,), generator=g, device=dev)).to(torch.uint8)
    strand = (torch.rand(n, generator=g, device=dev) < 0.5).to(torch.uint8)
    return s, e, mapq, strand


def synth_gaps(contig_len):
    """Synthetic centromere / telomere constants (hg19-like proportions)."""
    c0 = int(contig_len * 0.40) // _WINDOW * _WINDOW
    return (c0, c0 + 3_000_000, [(0, 10_000), (contig_len - 10_000, contig_len)])


def synth_blacklist(contig_len, seed, n_regions):
    rng = np.random.default_rng(seed)
    s = np.sort(rng.integers(0, contig_len - 6000, n_regions)).astype(np.int32)
    e = (s + rng.integers(200, 5000, n_regions)).astype(np.int32)
    order = np.lexsort((e, s))
    return s[order], e[order]
