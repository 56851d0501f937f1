// framer_kernels.hip -- the bit-domain back end of the receive path on the device:
// unique-word search, length / crc8 check and packet extraction over the demodulators' bit streams.
//
// Replaces RtlSdr_Receiver::deserializeData (RtlSdr_Receiver/rtlsdr_receiver.cpp:2197-2230) driving
// PacketFramer<T>::deserialize (RadioNode/src/libraries/network/PacketFramer.h:424-513) over a RingBuffer<T>
// (common/RingBuffer.h) with crc8::getCRC (crc/crc8.cpp:6-44): the stream of one call is the ring buffer's
// content, events come out in the order the reference would emit them, and `consumed` is where its front
// would stand, so that the caller carries bits[consumed..n) into the next call.
//
// One workgroup per channel.  The reference removes one bit at a time until the 16 front bits are the
// unique word; which bits it removes does not depend on its pass structure, so the search for the next
// header runs 2048 positions at a time across the workgroup.  What does depend on it -- a call returning
// 0 after 500 removed bits, or at the end of a pass once fewer than a packet's bits are left -- only
// matters within the last packet length of the stream, and that stretch is replayed literally by one lane.
#include "sdrx_internal.h"

namespace sdrx {

constexpr int PKT_BITS = 2032;          // MAX_PKT_SIZE_BITS = (250 + 4) * 8, PacketFramer.h:40-54
constexpr unsigned UW_ONES = 0x735Fu;   // PACKET_FRAMER_UNIQUE_WORD (PacketFramer.h:17-19), bit i = word[i]
constexpr int SEARCH_SPAN = 2048;       // positions examined per workgroup step (8 per lane)
constexpr int WIN_NEED = SEARCH_SPAN - 1 + PKT_BITS; // a search step at window offset w reads below w + WIN_NEED: the
                                                      // positions searched and a whole packet behind the last of them
constexpr int WIN_BYTES = 4 * 4352;      // the window kept in LDS: a step needs WIN_NEED + up to 15 bytes of alignment; four
                                         // times that, so that a stretch of packets (or of noise) is loaded once

__device__ __forceinline__ unsigned to_byte(const uint8_t *b) // PacketFramer.h:537-549: MSB first, == 1
{
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) c |= (b[i] == 1 ? 1u : 0u) << (7 - i);
    return c;
}

__global__ void __launch_bounds__(256) deframe_kernel(const uint8_t *__restrict__ bits_all, size_t n, size_t in_pitch,
                                                      int crc_enable, sdrx_frame_event *__restrict__ events_all,
                                                      uint8_t *__restrict__ payloads_all, size_t max_events,
                                                      uint64_t *__restrict__ summary_all)
{
    __shared__ uint8_t crc_table[256];
    __shared__ uint8_t crc_pow[251][8]; // crc_pow[k][b] = table applied k times to 1 << b
    __shared__ unsigned crc_part[4];
    __shared__ uint8_t pkt[256];
    __shared__ __align__(16) uint8_t win[WIN_BYTES]; // the stream from (just before) the search base on
    __shared__ int s_min, s_near;
    __shared__ int s_ok;
    __shared__ long long s_tail_front, s_tail_q;
    __shared__ unsigned long long s_tail_zero;
    const unsigned tid = threadIdx.x;
    const size_t chan = blockIdx.x;
    const uint8_t *bits = bits_all + chan * in_pitch;
    sdrx_frame_event *events = events_all + chan * max_events;
    uint8_t *payloads = payloads_all + chan * max_events * 250;

    { // crc8::crc8 / getCRCForByte (crc8.cpp:6-31), CRC7_POLY 0x91
        unsigned val = tid;
        for (int j = 0; j < 8; j++) {
            if (val & 1) val ^= 0x91;
            val >>= 1;
        }
        crc_table[tid] = (uint8_t)val;
    }
    __syncthreads();
    // getCRC (crc8.cpp:35-44) is crc = table[crc ^ m[i]] from 0, and the table is a LINEAR map over GF(2)
    // (shift and conditional xor of a constant on the low bit; table[0] = 0), so
    // crc(m[0..len)) = xor_i table^(len-i)(m[i]): one term per lane and a xor-reduction, not a 250-step chain.
    if (tid < 8) {
        unsigned v = 1u << tid;
        for (int k = 1; k <= 250; k++) {
            v = crc_table[v];
            crc_pow[k][tid] = (uint8_t)v;
        }
    }
    __syncthreads();

    long long p = 0; // the ring buffer's front
    unsigned long long nev = 0, zero_returns = 0;
    const long long N = (long long)n, lim = N - PKT_BITS; // fronts below lim: a whole packet and more is left

    long long wbase = -1; // win[0 .. wlen) = bits[wbase .. wbase + wlen) (2 beyond the stream's end)
    int wlen = 0;
    while (N - p > PKT_BITS) { // the receiver's loop condition, at the start of a deserialize call
        // ---- first front q in [p, lim) whose 16 bits are the unique word
        long long q = -1;
        for (long long base = p; base < lim && q < 0; base += SEARCH_SPAN) {
            bool refilled = false;
            if (wbase < 0 || base < wbase || base - wbase + WIN_NEED > wlen) { // (uniform) this step leaves the window
                // the window starts at the 16-byte boundary at or below `base`: 16-byte loads
                long long start = base - (long long)((size_t)(bits + base) & 15);
                if (start < 0) start = base; // (the channel's first bytes: no reading in front of the buffer)
                const bool vec = (((size_t)(bits + start)) & 15) == 0;
                for (int e = (int)tid * 16; e < WIN_BYTES; e += 256 * 16) {
                    if (vec && start + e + 16 <= N) {
                        *(uint4 *)(win + e) = *(const uint4 *)(bits + start + e);
                    } else {
                        for (int k = 0; k < 16; k++) win[e + k] = (start + e + k < N) ? bits[start + e + k] : (uint8_t)2;
                    }
                }
                wbase = start;
                wlen = WIN_BYTES;
                refilled = true;
            }
            const int woff = (int)(base - wbase);
            if (refilled) __syncthreads();
            if (base == p) { // the front itself -- a packet right behind the last one: every wave looks for itself (the
                const unsigned lane = tid & 63; // window is the same for all), no barrier, no search
                const bool ok = lane >= 16 || win[woff + lane] == ((UW_ONES >> lane) & 1u);
                if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) {
                    q = base;
                    break;
                }
                // the next 512 positions by the first wave alone (8 per lane; the lowest lane with a hit holds the lowest
                // position): one barrier, and the other waves issue nothing -- the kernel is bound by the instructions
                // a workgroup issues per event, and in a busy stream the next header is seldom far
                if (tid < 64) {
                    unsigned ones = 0, zeros = 0;
#pragma unroll
                    for (int k = 0; k < 23; k++) {
                        const unsigned v = win[woff + 8 * tid + k];
                        ones |= (v == 1 ? 1u : 0u) << k;
                        zeros |= (v == 0 ? 1u : 0u) << k;
                    }
                    int first = 0x7fffffff;
#pragma unroll
                    for (int o = 7; o >= 0; o--) {
                        const bool hit = ((ones >> o) & 0xffffu) == UW_ONES && ((zeros >> o) & 0xffffu) == (~UW_ONES & 0xffffu);
                        if (hit && base + 8 * (long long)tid + o < lim) first = 8 * (int)tid + o;
                    }
                    const unsigned long long any = __builtin_amdgcn_ballot_w64(first != 0x7fffffff);
                    const int best = any ? __builtin_amdgcn_readlane(first, (int)__builtin_ctzll(any)) : 0x7fffffff;
                    if (tid == 0) s_near = best;
                }
                __syncthreads();
                const int near = s_near; // (written again only after the barriers that end this event)
                if (near != 0x7fffffff) {
                    q = base + near;
                    break;
                }
            }
            if (tid == 0) s_min = 0x7fffffff;
            __syncthreads();
            unsigned ones = 0, zeros = 0;
#pragma unroll
            for (int k = 0; k < 23; k++) {
                const unsigned v = win[woff + 8 * tid + k];
                ones |= (v == 1 ? 1u : 0u) << k;
                zeros |= (v == 0 ? 1u : 0u) << k;
            }
            int first = 0x7fffffff;
#pragma unroll
            for (int o = 7; o >= 0; o--) {
                const bool hit = ((ones >> o) & 0xffffu) == UW_ONES && ((zeros >> o) & 0xffffu) == (~UW_ONES & 0xffffu);
                if (hit && base + 8 * (long long)tid + o < lim) first = 8 * (int)tid + o;
            }
            if (first != 0x7fffffff) atomicMin(&s_min, first);
            __syncthreads();
            if (s_min != 0x7fffffff) q = base + s_min;
            __syncthreads();
        }

        bool last_call = false;
        if (q >= 0) {
            zero_returns += (unsigned long long)(q - p) / 500; // a call gives up after 500 removed bits; the next goes on
        } else {
            // ---- no header before lim: replay the reference literally over the end of the stream.
            // Up to f0 the simple picture holds; there the state is known: a removal leaves the pass at
            // index 1 with validHeader false, unless it was a call's 500th (then a fresh call starts).
            if (tid == 0) {
                long long f0 = lim - 600 > p ? lim - 600 : p;
                const long long r = f0 - p;
                unsigned long long zr = (unsigned long long)r / 500;
                long long front = f0, tq = -1;
                int bad = (int)(r % 500);
                bool vh = (bad == 0);
                int i = vh ? 0 : 1;
                for (;;) { // calls
                    bool accepted = false, returned = false;
                    for (;;) { // passes of one call (PacketFramer.h:444-469)
                        while (i < 16) {
                            if (bits[front + i] != ((UW_ONES >> i) & 1u)) {
                                bad++;
                                front++;
                                vh = false;
                                i = 1;
                                if (bad >= 500) { returned = true; break; }
                            } else {
                                i++;
                            }
                        }
                        if (returned) break;
                        if (N - front < PKT_BITS) { returned = true; break; }
                        if (vh) { accepted = true; break; }
                        vh = true;
                        i = 0;
                    }
                    if (accepted) { tq = front; break; }
                    zr++; // the call returned 0
                    if (N - front > PKT_BITS) { bad = 0; vh = true; i = 0; continue; } // the receiver calls again
                    break;
                }
                s_tail_front = front;
                s_tail_q = tq;
                s_tail_zero = zr;
            }
            __syncthreads();
            zero_returns += s_tail_zero;
            q = s_tail_q;
            p = s_tail_front;
            __syncthreads();
            if (q < 0) break; // the stream ran out: p is the final front
            last_call = true;   // a header exactly one packet before the end: validate it, then stop
        }

        // ---- header at q, at least PKT_BITS bits from there: length, crc, data (PacketFramer.h:474-512)
        if (wbase < 0 || q < wbase || q + PKT_BITS > wbase + wlen) { // (only after the literal replay)
            for (int e = (int)tid; e < PKT_BITS; e += 256) win[e] = bits[q + e];
            wbase = q;
            wlen = PKT_BITS;
            __syncthreads();
        }
        const uint8_t *hdr = win + (q - wbase);
        // length and crc bytes (to_byte: MSB first) out of one ballot per wave over the header's first 64 bits
        const unsigned long long hbits = __builtin_amdgcn_ballot_w64(hdr[tid & 63] == 1);
        const unsigned len = __brev((unsigned)(hbits >> 16) & 0xffu) >> 24;
        unsigned crc = 0;
        int kind;
        if (len > 250) {
            kind = 2;
            p = q + 1;
            __syncthreads(); // every wave has read the header before the next search refills the window
        } else {
            crc = __brev((unsigned)(hbits >> 24) & 0xffu) >> 24;
            if (tid < 250) pkt[tid] = (uint8_t)to_byte(hdr + 32 + 8 * tid);
            __syncthreads();
            unsigned term = 0;
            if (tid < len) {
                const unsigned m = pkt[tid];
                const uint8_t *col = crc_pow[len - tid];
#pragma unroll
                for (int b = 0; b < 8; b++) term ^= ((m >> b) & 1u) ? col[b] : 0u;
            }
#pragma unroll
            for (int sh = 32; sh >= 1; sh >>= 1) term ^= __shfl_xor(term, sh);
            if ((tid & 63) == 0) crc_part[tid >> 6] = term;
            __syncthreads();
            if (tid == 0) {
                const unsigned c = crc_part[0] ^ crc_part[1] ^ crc_part[2] ^ crc_part[3];
                s_ok = (!crc_enable || c == crc) ? 1 : 0;
            }
            __syncthreads();
            kind = s_ok ? 1 : 3;
            if (kind == 1 && nev < max_events && tid < 250) payloads[nev * 250 + tid] = pkt[tid];
            p = kind == 1 ? q + PKT_BITS : q + 1;
            __syncthreads(); // pkt and s_ok are reused by the next event
        }
        if (tid == 0 && nev < max_events) {
            sdrx_frame_event ev;
            ev.pos = q;
            ev.kind = kind;
            ev.len = (uint8_t)len;
            ev.crc = (uint8_t)crc;
            ev.pad[0] = ev.pad[1] = 0;
            events[nev] = ev;
        }
        nev++;
        if (last_call) break;
    }
    if (tid == 0) {
        uint64_t *s = summary_all + 4 * chan;
        s[0] = nev;
        s[1] = (uint64_t)p;
        s[2] = zero_returns;
        s[3] = 0;
    }
}

int launch_deframe(const uint8_t *bits, size_t n, size_t nchan, size_t in_pitch, int crc_enable, sdrx_frame_event *events,
                   uint8_t *payloads, size_t max_events, uint64_t *summary, hipStream_t s)
{
    if (!summary || (nchan && n && !bits) || (max_events && (!events || !payloads))) {
        set_error("sdrx_deframe_dev: null pointer");
        return SDRX_EINVAL;
    }
    if (nchan == 0) return SDRX_OK;
    if (nchan > 0x7fffffffULL || n > 0x3fffffffffffULL) return SDRX_EUNSUPPORTED;
    hipLaunchKernelGGL(deframe_kernel, dim3((unsigned)nchan), dim3(256), 0, s, bits, n, in_pitch, crc_enable, events,
                       payloads, max_events, summary);
    SDRX_HIP(hipGetLastError());
    return SDRX_OK;
}

} // namespace sdrx
