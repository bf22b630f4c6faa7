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

// Enqueue the whole parse of text[0, n) (complete lines, the last byte is '\n') on `s`.
//   d_block_count: ceil(n / kTextBlockBytes) + 1 words of scratch; d_line_start: max_lines + 2 words;
//   outputs hold max_lines rows; *d_sum must be zeroed by the caller (stream-ordered) beforehand.
void textparse_launch(hipStream_t s, const uint8_t* d_text, size_t n, bool bed6, uint32_t* d_block_count,
                      uint32_t* d_line_start, size_t max_lines, int32_t* d_start, int32_t* d_end, uint8_t* d_mapq,
                      uint8_t* d_strand, TextSummary* d_sum);

// The same for a piece whose text was INFLATED ON THE DEVICE into d_text[data_off, data_off + data_len) (data_off >=
// kTextCarryMax): a set-up kernel first moves the unfinished last line of the previous piece (prev_text / prev_sum,
// NULL for the first piece) in front of the data, skips first_skip bytes (first piece after an index seek), finds
// the last line end -- at the end of the file a missing one is appended -- and records the range in *d_sum
// (text_off, text_len, tail_len); the parse kernels then take the range from there.  Run names go to
// d_sum->run_name.  *d_sum must be zeroed beforehand.  run_off values are relative to text_off & ~15.
void textparse_launch_inflated(hipStream_t s, uint8_t* d_text, uint32_t data_off, uint32_t data_len,
                               const uint8_t* prev_text, const TextSummary* prev_sum, uint32_t first_skip, bool eof, bool bed6,
                               uint32_t* d_block_count, uint32_t* d_line_start, size_t max_lines, int32_t* d_start,
                               int32_t* d_end, uint8_t* d_mapq, uint8_t* d_strand, TextSummary* d_sum);

}  // namespace ftk
