// DEFLATE (RFC 1951) on the device for BGZF files: one wavefront per BGZF block.
//
// The streaming text decoder spends most of its wall time in the host's inflate (33 of 52 ms for a four-contig
// fragment file on 16 threads, profiles/r2_*): BGZF blocks are independent DEFLATE streams of at most 64 KB of
// data, thousands per file piece, which is the parallelism a GPU wants -- the serial part (Huffman decoding, one
// symbol after the other) stays serial inside a block and runs as uniform (scalar) work of one wave; the wave's
// 64 lanes do the table construction, the LZ77 copies and the stores.  See ftk_inflate.hip.
#pragma once
#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime_api.h>

namespace ftk {

struct InflateBlock {
    uint32_t in_off;   // first byte of the raw DEFLATE payload in the compressed buffer
    uint32_t in_len;   // payload bytes
    uint32_t out_off;  // where the block's data goes in the output buffer
    uint32_t out_len;  // ISIZE of the block
};

struct InflateStatus {
    unsigned int n_bad;      // blocks that did not decode to exactly out_len bytes
    unsigned int first_bad;  // index of one of them (valid when n_bad > 0)
    unsigned int reason;     // what went wrong there (kInflate* code)
};

enum : unsigned {
    kInflateOk = 0,
    kInflateBadBlockType = 1,
    kInflateBadStored = 2,
    kInflateBadLengths = 3,   // code lengths do not describe a prefix code / repeat without a previous length
    kInflateBadSymbol = 4,    // a bit pattern that is no code, or a length / distance symbol out of range
    kInflateBadDistance = 5,  // distance reaches in front of the block's data
    kInflateOverrun = 6,      // more data than ISIZE, or the payload ran out
    kInflateShort = 7,        // final block ended before ISIZE bytes
};

// Enqueue the inflate of n_blocks BGZF payloads on `s`: d_comp holds the compressed bytes (readable up to the
// next multiple of 4 behind the last payload), d_out receives the data (base aligned to 4 KB), *d_status must be
// zeroed beforehand (stream-ordered).  d_crc (may be NULL): CRC-32 of every block's data.  vector_matches: resolve a
// window's matches on the lanes side by side instead of one after the other - pays for BAM records (few, far-apart
// matches among literals: -5 %), not for fragment rows (+1..6 %): ftk_inflate.hip.
// the lane-parallel loop's scratch (per device, ~370 MB) back to the device; bytes freed.  Only with no inflate launch in flight.
size_t inflate_release_scratch();
void inflate_launch(hipStream_t s, const uint8_t* d_comp, const InflateBlock* d_tab, int n_blocks, uint8_t* d_out,
                    InflateStatus* d_status, uint32_t* d_crc, bool vector_matches = false);

}  // namespace ftk
