// Device-side row parser of the streaming text decoder (a2: io/alignment.py:270-302).
// The host inflates a piece of the frag.gz into page-locked memory; the GPU finds the lines and parses
// the plain rows (name, unsigned decimals, one strand character) into fragment columns.  A piece with
// anything else in it is parsed by the host's field-rule parser instead (ftk_decode.cpp), so the result
// is the same either way.
#pragma once
#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime_api.h>

namespace ftk {

constexpr int kTextMaxRuns = 1024;      // contig runs reported per piece (more: the host parses the piece)
constexpr int kTextBlockBytes = 4096;   // bytes per block of the newline kernels

struct TextSummary {
    unsigned long long n_lines;   // '\n' count of the piece
    unsigned long long n_bad;     // lines that are not plain accepted rows
    unsigned int n_runs;          // lines whose contig name differs from the previous line's (line 0 included)
    unsigned int overflow;        // more lines than the output arrays hold
    unsigned int run_line[kTextMaxRuns];
    unsigned int run_off[kTextMaxRuns];   // byte offset of that line in the piece
};

// Enqueue the whole parse of text[0, n) (complete lines, the last byte is '\n') on `s`.
//   d_block_count: ceil(n / kTextBlockBytes) + 1 words of scratch; d_line_start: max_lines + 2 words;
//   outputs hold max_lines rows; *d_sum must be zeroed by the caller (stream-ordered) beforehand.
void textparse_launch(hipStream_t s, const uint8_t* d_text, size_t n, bool bed6, uint32_t* d_block_count,
                      uint32_t* d_line_start, size_t max_lines, int32_t* d_start, int32_t* d_end, uint8_t* d_mapq,
                      uint8_t* d_strand, TextSummary* d_sum);

}  // namespace ftk
