// One BAM alignment record -> one fragment row: the ONE statement of the reference's BAM rule in this library
// (io/alignment.py:60-71 `_read_is_low_quality`, :242-268 `_fetch_sam`), shared by the device record parser
// (ftk_bamparse.hip) and the two host decoders (ftk_decode.cpp).  Pinned to the reference's own code by
// tests/golden/bam.json.gz (vectors the imported reference produced over a BAM stand-in for pysam).
//
// What pysam / htslib contribute to the reference's result and is restated here:
//   * reference_end (pysam libcalignedsegment.pyx): None when the read has no CIGAR, else htslib's bam_endpos;
//   * bam_endpos (htslib sam.c): pos + reference length of the CIGAR (ops M D N = X), an alignment that consumes no
//     reference counting as ONE base - this is also the end htslib's region iterator tests (`fetch`), CIGAR or not.
#pragma once
#include <stdint.h>

#if defined(__HIP__) || defined(__HIPCC__)
#define FTK_BAMRULE_HD __host__ __device__ __forceinline__
#else
#define FTK_BAMRULE_HD inline
#endif

namespace ftk {

struct BamRow {
    int32_t fs, fe;    // the fragment [fs, fe)
    int32_t r1s, r1e;  // read1's alignment [pos, bam_endpos): what a region query tests (io/alignment.py:245)
    uint8_t mapq, fwd;
};

enum BamRule : int {
    kBamNotFragment = 0,  // filtered by flag, read2, TLEN 0, or a record whose CIGAR does not fit its block_size
    kBamFragment = 1,
    // the reference yields a fragment the columns cannot hold: a negative start (reference_end + TLEN < 0) or a
    // coordinate beyond int32.  Dropped and COUNTED (ftk_fragstream_skipped / ftk_fragtable_skipped): the Python
    // surface warns, the reference would have kept the fragment.
    kBamUnrepresentable = 2,
    // read1 without a CIGAR and TLEN < 0: the reference evaluates `None + tlen` (io/alignment.py:257) and raises
    // TypeError.  Dropped and counted; the Python surface raises TypeError like the reference.
    kBamNoCigarReverse = 3,
};

// r: the record behind its block_size field (bs bytes).  RD32 reads a little-endian uint32 at an unaligned address.
template <class RD32>
FTK_BAMRULE_HD int bam_rule(const uint8_t* r, uint32_t bs, RD32 rd32, BamRow& f) {
    const int32_t pos = (int32_t)rd32(r + 4);
    const uint32_t w8 = rd32(r + 8);    // l_read_name | mapq << 8 | bin << 16
    const uint32_t w12 = rd32(r + 12);  // n_cigar_op | flag << 16
    const uint32_t l_read_name = w8 & 0xffu, n_cigar = w12 & 0xffffu, flag = w12 >> 16;
    const int32_t tlen = (int32_t)rd32(r + 28);
    // _read_is_low_quality (:60-71) without the mapq cut, which the kernels apply: unmapped, secondary, not paired,
    // mate unmapped, duplicate, qc-fail, supplementary, not a proper pair
    if ((flag & 0x4u) || (flag & 0x100u) || !(flag & 0x1u) || (flag & 0x8u) || (flag & 0x400u) || (flag & 0x200u) ||
        (flag & 0x800u) || !(flag & 0x2u))
        return kBamNotFragment;
    if (flag & 0x80u) return kBamNotFragment;  // read1_only (:248): read2 is skipped
    if (tlen == 0) return kBamNotFragment;     // (:259-260)
    if (32ull + l_read_name + 4ull * n_cigar > bs) return kBamNotFragment;
    const uint8_t* cg = r + 32 + l_read_name;
    long long ref_len = 0;
    for (uint32_t k = 0; k < n_cigar; ++k) {
        const uint32_t v = rd32(cg + 4 * k);
        const uint32_t op = v & 15u;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += v >> 4;
    }
    const long long end_pos = (long long)pos + (ref_len ? ref_len : 1);  // bam_endpos
    long long fs, fe;
    if (tlen > 0) {  // (:253-255) no reference_end needed: a read without a CIGAR counts too
        fs = pos;
        fe = (long long)pos + tlen;
    } else {         // (:256-258) reference_end + tlen .. reference_end
        if (n_cigar == 0) return kBamNoCigarReverse;
        fs = end_pos + tlen;
        fe = end_pos;
    }
    if (fs < 0 || fe < 0 || fs > INT32_MAX || fe > INT32_MAX || end_pos > INT32_MAX) return kBamUnrepresentable;
    f.fs = (int32_t)fs;
    f.fe = (int32_t)fe;
    f.r1s = pos;
    f.r1e = (int32_t)end_pos;
    f.mapq = (uint8_t)((w8 >> 8) & 0xffu);
    f.fwd = (flag & 0x10u) ? 0 : 1;  // is_forward (:266)
    return kBamFragment;
}

}  // namespace ftk
