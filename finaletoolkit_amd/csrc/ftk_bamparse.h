// Device-side BAM record parser of the streaming decoder (a3: io/alignment.py:60-71,242-268).
// After bgzf_inflate_kernel the records of a piece lie in HBM; these kernels find the record chain, apply the
// reference's read filter and fragment reconstruction, and leave fragment columns (start, end, mapq, strand, read1
// span, reference id) in file order on the device - the inflated bytes never cross PCIe.  A contig's rows are then
// sorted by fragment start on the device (bam_sort_contig).  The host decoder (ftk_decode.cpp: run_bam's walk /
// bam_record) states the same rules and stays as the fallback for a piece whose chain the stretches cannot settle.
#pragma once
#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime_api.h>

namespace ftk {

constexpr int kBamMaxRuns = 64;           // contig runs reported per piece (more: the host walks the piece)
constexpr uint32_t kBamNoStart = 0xffffffffu;

struct BamSummary {
    uint32_t base_off;        // where the chain's bytes start in the slot's buffer: the carry sits in front of the data
    uint32_t m;               // bytes of the chain's range [base_off, base_off + m): carry + this piece's data
    uint32_t landing;         // offset in that range where the chain stops: the first incomplete record (or m)
    uint32_t consistent;      // 1: every stretch starts where the one before it landed - the columns are the chain's
    uint32_t bad;             // a record with block_size < 32 on the chain
    uint32_t carry_overflow;  // the previous piece's unfinished record does not fit in front of the data
    uint32_t n_rows;          // fragments written
    uint32_t n_records;       // records on the chain, any reference
    uint32_t n_unrepresentable;  // records the reference makes a fragment of that the columns cannot hold (ftk_bamrule.h)
    uint32_t n_nocigar_reverse;  // CIGAR-less read1 records with TLEN < 0: the reference raises TypeError on them
    uint32_t n_runs;          // contig runs among the fragments (> kBamMaxRuns: too many to list)
    uint32_t n_stretch;       // stretches the range was cut into
    uint32_t n_repairs;       // stretches the last kernel had to walk again serially
    uint32_t first_unsettled; // (when not consistent) the first stretch that does not start where its predecessor landed
    uint32_t dbg[4];          // its start, the predecessor's landing, the predecessor's start, the landing before that
    unsigned long long last_key;  // (reference id as uint32) << 32 | position of the LAST record on the chain in this piece
                                  // (0: no record; unmapped reads at the end of the file carry reference 0xffffffff):
                                  // region reads stop when it passes the region's end
    uint32_t run_row[kBamMaxRuns];
    int32_t run_ref[kBamMaxRuns];
};

// Scratch per piece: 4 words per stretch (start, landing, count, row offset).
inline size_t bam_stretch_words(size_t max_bytes, uint32_t stretch_bytes) { return 4 * (max_bytes / stretch_bytes + 2); }

// Enqueue the whole parse of one piece on `s`.  The piece's inflated bytes are d_text[data_off, data_off + data_len);
// the unfinished record the previous piece ended in (prev_text / prev_sum: that piece's buffer and summary, both on the
// device; NULL for the first piece) is moved in front of them first.  first_off: bytes of the data to skip before the
// chain starts (the BAM header in the first piece; the offset an index seek points to).  wanted[ref] != 0 keeps a
// reference's records.  Outputs hold max_rows rows; *d_sum must be zeroed (stream-ordered) beforehand.
void bamparse_launch(hipStream_t s, uint8_t* d_text, uint32_t data_off, uint32_t data_len, const uint8_t* prev_text,
                     const BamSummary* prev_sum, uint32_t first_off, const uint8_t* d_wanted, int n_ref,
                     uint32_t stretch_bytes, uint32_t* d_stretch, size_t stretch_words, size_t max_rows, int32_t* d_start,
                     int32_t* d_end, uint8_t* d_mapq, uint8_t* d_strand, int32_t* d_r1s, int32_t* d_r1e, int32_t* d_ref,
                     BamSummary* d_sum);

// A contig's rows (file order: n rows in the in_* columns) -> fragment-start order, stable (ties keep file order):
// out_* columns and out_ord[i] = file rank of row i.  d_tmp: bam_sort_tmp_bytes(n) bytes of scratch.
size_t bam_sort_tmp_bytes(size_t n);
int bam_sort_contig(hipStream_t s, size_t n, const int32_t* in_start, const int32_t* in_end, const uint8_t* in_mapq,
                    const uint8_t* in_strand, const int32_t* in_r1s, const int32_t* in_r1e, int32_t* out_start,
                    int32_t* out_end, uint8_t* out_mapq, uint8_t* out_strand, int32_t* out_r1s, int32_t* out_r1e,
                    int32_t* out_ord, void* d_tmp, size_t tmp_bytes);

}  // namespace ftk
