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
constexpr int kTextBlockBytes = 4096;   // bytes of text per block of the line finder

constexpr int kTextNamedRuns = 64;      // runs whose contig name the device reports itself (device-inflated pieces)
constexpr int kTextNameBytes = 48;      // longest such name + 1
constexpr unsigned kTextCarryMax = 1u << 16;  // longest unfinished line carried from one piece to the next on the device

struct TextSummary {
    unsigned long long n_lines;   // '\n' count of the piece
    unsigned long long n_bad;     // lines that are not plain accepted rows
    unsigned int n_runs;          // lines whose contig name differs from the previous line's (line 0 included)
    unsigned int overflow;        // more lines than the output arrays hold
    unsigned int run_line[kTextMaxRuns];
    unsigned int run_off[kTextMaxRuns];   // byte offset of that line in the piece
    // device-inflated pieces (textparse_launch_inflated): where the piece's complete lines lie in the set's text
    // buffer, what is left behind them, and the names of the first runs
    unsigned int text_off, text_len, tail_len;
    unsigned int name_overflow;   // a run beyond kTextNamedRuns or a name that does not fit: the host reads the text
    unsigned int carry_overflow;  // an unfinished line longer than kTextCarryMax
    unsigned int last_line_bad;   // the piece's LAST line is one of the n_bad (a row cut off by an index-driven read
                                  // that stops inside a block: the host may ignore exactly that one)
    unsigned char run_name[kTextNamedRuns][kTextNameBytes];  // NUL-terminated, indexed like run_line / run_off
};

// Device scratch the parse of a piece of `text_bytes` of text needs (the line finder's scan state).
size_t textparse_scratch_bytes(size_t text_bytes);

// Enqueue the whole parse of text[0, n) (complete lines, the last byte is '\n') on `s`.
//   d_scratch: textparse_scratch_bytes(n); outputs hold max_lines rows; *d_sum must be zeroed by the caller
//   (stream-ordered) beforehand.
void textparse_launch(hipStream_t s, const uint8_t* d_text, size_t n, bool bed6, void* d_scratch, size_t max_lines,
                      int32_t* d_start, int32_t* d_end, uint8_t* d_mapq, uint8_t* d_strand, TextSummary* d_sum);

// The same for a piece whose text was INFLATED ON THE DEVICE into d_text[data_off, data_off + data_len) (data_off >=
// kTextCarryMax): a set-up kernel first moves the unfinished last line of the previous piece (prev_text / prev_sum,
// NULL for the first piece) in front of the data, skips first_skip bytes (first piece after an index seek), finds
// the last line end -- at the end of the file a missing one is appended -- and records the range in *d_sum
// (text_off, text_len, tail_len); the parse kernel then takes the range from there.  Run names go to
// d_sum->run_name.  *d_sum must be zeroed beforehand.  run_off values are relative to text_off & ~15.
// d_scratch: textparse_scratch_bytes(data_len).
void textparse_launch_inflated(hipStream_t s, uint8_t* d_text, uint32_t data_off, uint32_t data_len,
                               const uint8_t* prev_text, const TextSummary* prev_sum, uint32_t first_skip, bool eof, bool bed6,
                               void* d_scratch, size_t max_lines, int32_t* d_start, int32_t* d_end, uint8_t* d_mapq,
                               uint8_t* d_strand, TextSummary* d_sum);

// Rows of a parsed piece behind a contig's columns, device to device, in ONE launch: n rows of (start, end, mapq,
// strand) and - BAM contigs - (r1_start, r1_end; NULL otherwise) from the piece's buffers to the contig's block at its
// current row count.  Neither side is aligned beyond its element size (a contig run starts at any row of a piece and
// lands at any row of the block).  Replaces four to six hipMemcpyAsync per run: the runtime's copy kernel moved the
// genome's 6 GB of appends at 0.15 TB/s, 15 % of the file -> HBM leg's GPU time (profiles/r3_g_genome_leg_kernel_stats.txt).
void append_rows_launch(hipStream_t s, int32_t* dst_start, int32_t* dst_end, uint8_t* dst_mapq, uint8_t* dst_strand,
                        int32_t* dst_r1s, int32_t* dst_r1e, const int32_t* src_start, const int32_t* src_end,
                        const uint8_t* src_mapq, const uint8_t* src_strand, const int32_t* src_r1s, const int32_t* src_r1e,
                        size_t n);

}  // namespace ftk
