// Motion-JPEG front end (`_init_video`, marker_detection.py:50-76 - the reference reads its AVI through
// cv2.VideoCapture, native code; decode is outside the metric but bounds what a recorded session sees).
//   host:   baseline JPEG entropy decode (Huffman, 8-bit, 1 or 3 components, 4:4:4 / 4:2:2 / 4:2:0, restart intervals) of a
//           batch of frames on plain C++ threads - no Python, no GIL - into quantised coefficient blocks in natural order
//   device: dequantisation + the 8x8 inverse DCT + chroma upsampling + YCbCr -> BGR, restated from the published libjpeg
//           algorithms (jidctint.c "islow", jdsample.c "fancy" triangle upsampling, jdcolor.c fixed-point tables) so that the
//           frames equal Pillow's decode (libjpeg-turbo) bit for bit: tests/test_gpu_parity.py::test_mjpeg_device_decode_*
// Whatever this does not take (progressive, arithmetic, 12-bit, other sampling) is reported and the caller decodes with Pillow.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/vbs.h"

namespace {

struct Huff {
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    // canonical decode tables
    int32_t maxcode[18];
    int32_t valptr[17];
    uint16_t mincode[17];
    // 9-bit look-ahead: (length << 8) | symbol, 0 = longer than 9 bits
    uint16_t look[512];
    // AC tables: a code AND the value bits behind it, where both fit the 9 bits: value << 16 | run << 8 | bits consumed
    // (0 = not this way); most coefficients of a camera frame are such short ones
    int32_t fast[512];
    bool ok = false;
    void build() {
        int k = 0, code = 0;
        uint16_t huffcode[257];
        uint8_t huffsize[257];
        for (int l = 1; l <= 16; ++l)
            for (int i = 0; i < bits[l]; ++i) huffsize[k++] = (uint8_t)l;
        huffsize[k] = 0;
        const int n = k;
        k = 0;
        int si = huffsize[0];
        bool valid = true;
        while (huffsize[k]) {
            while (huffsize[k] == si) huffcode[k++] = (uint16_t)code++;
            if (code > (1 << si)) valid = false;        // more codes of this length than there are (a corrupt DHT): refused,
            code <<= 1;                                  // as libjpeg does - the look-ahead table below is indexed by the codes
            ++si;
        }
        if (!valid) { ok = false; return; }
        int p = 0;
        for (int l = 1; l <= 16; ++l) {
            if (bits[l]) {
                valptr[l] = p;
                mincode[l] = huffcode[p];
                p += bits[l];
                maxcode[l] = huffcode[p - 1];
            } else {
                maxcode[l] = -1;
                valptr[l] = 0;
                mincode[l] = 0;
            }
        }
        maxcode[17] = 0xFFFFF;
        memset(look, 0, sizeof look);
        p = 0;
        for (int l = 1; l <= 9; ++l)
            for (int i = 0; i < bits[l]; ++i, ++p) {
                const int first = huffcode[p] << (9 - l);
                for (int c = 0; c < (1 << (9 - l)); ++c) look[first + c] = (uint16_t)((l << 8) | vals[p]);
            }
        for (int i = 0; i < 512; ++i) {
            fast[i] = 0;
            const int e = look[i];
            if (!e) continue;
            const int len = e >> 8, sym = e & 255, run = sym >> 4, sz = sym & 15;
            if (sz == 0 || len + sz > 9) continue;
            int v = (i >> (9 - len - sz)) & ((1 << sz) - 1);
            if (v < (1 << (sz - 1))) v -= (1 << sz) - 1;
            fast[i] = (int32_t)((uint32_t)v << 16 | (uint32_t)run << 8 | (uint32_t)(len + sz));
        }
        ok = n > 0;
    }
};

struct Frame {
    int w = 0, h = 0, ncomp = 0;
    int hs[3] = {1, 1, 1}, vs[3] = {1, 1, 1}, tq[3] = {0, 0, 0}, td[3] = {0, 0, 0}, ta[3] = {0, 0, 0}, id[3] = {0, 0, 0};
    uint16_t qt[4][64];
    bool qt_ok[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    int restart = 0;
    const uint8_t* scan = nullptr;
    const uint8_t* end = nullptr;
};

const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                            41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                            30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

// ITU-T T.81 Annex K.3 "typical" Huffman tables: what an encoder that does not optimise writes, and what a Motion-JPEG frame
// WITHOUT a DHT segment means (AVI MJPG from cameras; libjpeg-turbo and FFmpeg install the same tables for such frames).
const uint8_t STD_DC_BITS[2][16] = {{0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0}, {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0}};
const uint8_t STD_AC_BITS[2][16] = {{0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125}, {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 119}};
const uint8_t STD_AC_VALS[2][162] = {
    {1,   2,   3,   0,   4,   17,  5,   18,  33,  49,  65,  6,   19,  81,  97,  7,   34,  113, 20,  50,  129, 145, 161, 8,   35,  66,  177,
     193, 21,  82,  209, 240, 36,  51,  98,  114, 130, 9,   10,  22,  23,  24,  25,  26,  37,  38,  39,  40,  41,  42,  52,  53,  54,  55,
     56,  57,  58,  67,  68,  69,  70,  71,  72,  73,  74,  83,  84,  85,  86,  87,  88,  89,  90,  99,  100, 101, 102, 103, 104, 105, 106,
     115, 116, 117, 118, 119, 120, 121, 122, 131, 132, 133, 134, 135, 136, 137, 138, 146, 147, 148, 149, 150, 151, 152, 153, 154, 162, 163,
     164, 165, 166, 167, 168, 169, 170, 178, 179, 180, 181, 182, 183, 184, 185, 186, 194, 195, 196, 197, 198, 199, 200, 201, 202, 210, 211,
     212, 213, 214, 215, 216, 217, 218, 225, 226, 227, 228, 229, 230, 231, 232, 233, 234, 241, 242, 243, 244, 245, 246, 247, 248, 249, 250},
    {0,   1,   2,   3,   17,  4,   5,   33,  49,  6,   18,  65,  81,  7,   97,  113, 19,  34,  50,  129, 8,   20,  66,  145, 161, 177, 193,
     9,   35,  51,  82,  240, 21,  98,  114, 209, 10,  22,  36,  52,  225, 37,  241, 23,  24,  25,  26,  38,  39,  40,  41,  42,  53,  54,
     55,  56,  57,  58,  67,  68,  69,  70,  71,  72,  73,  74,  83,  84,  85,  86,  87,  88,  89,  90,  99,  100, 101, 102, 103, 104, 105,
     106, 115, 116, 117, 118, 119, 120, 121, 122, 130, 131, 132, 133, 134, 135, 136, 137, 138, 146, 147, 148, 149, 150, 151, 152, 153, 154,
     162, 163, 164, 165, 166, 167, 168, 169, 170, 178, 179, 180, 181, 182, 183, 184, 185, 186, 194, 195, 196, 197, 198, 199, 200, 201, 202,
     210, 211, 212, 213, 214, 215, 216, 217, 218, 226, 227, 228, 229, 230, 231, 232, 233, 234, 242, 243, 244, 245, 246, 247, 248, 249, 250}};

struct StdTables {
    Huff dc[2], ac[2];
    StdTables() {
        for (int k = 0; k < 2; ++k) {
            int n = 0;
            for (int i = 0; i < 16; ++i) { dc[k].bits[i + 1] = STD_DC_BITS[k][i]; n += STD_DC_BITS[k][i]; }
            for (int i = 0; i < n; ++i) dc[k].vals[i] = (uint8_t)i;
            dc[k].build();
            for (int i = 0; i < 16; ++i) ac[k].bits[i + 1] = STD_AC_BITS[k][i];
            memcpy(ac[k].vals, STD_AC_VALS[k], 162);
            ac[k].build();
        }
    }
};
const StdTables& std_tables() { static const StdTables t; return t; }

// headers up to and including SOS; VBS_OK, or VBS_EINVAL for a stream this decoder does not take
int parse(const uint8_t* d, int64_t size, Frame* f) {
    if (size < 4 || d[0] != 0xFF || d[1] != 0xD8) return VBS_EINVAL;
    int64_t p = 2;
    while (p + 4 <= size) {
        if (d[p] != 0xFF) return VBS_EINVAL;
        const int m = d[p + 1];
        if (m == 0xFF) { ++p; continue; }               // fill byte
        p += 2;
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (p + 2 > size) return VBS_EINVAL;
        const int len = be16(d + p);
        if (len < 2 || p + len > size) return VBS_EINVAL;
        const uint8_t* s = d + p + 2;
        const int n = len - 2;
        if (m == 0xC0 || m == 0xC1) {                    // baseline / extended sequential, Huffman
            if (n < 6 || s[0] != 8) return VBS_EINVAL;
            f->h = be16(s + 1); f->w = be16(s + 3); f->ncomp = s[5];
            if ((f->ncomp != 1 && f->ncomp != 3) || n < 6 + 3 * f->ncomp || f->w < 1 || f->h < 1) return VBS_EINVAL;
            for (int c = 0; c < f->ncomp; ++c) {
                f->id[c] = s[6 + 3 * c]; f->hs[c] = s[7 + 3 * c] >> 4; f->vs[c] = s[7 + 3 * c] & 15; f->tq[c] = s[8 + 3 * c] & 3;
            }
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return VBS_EINVAL;                           // progressive, lossless, arithmetic
        } else if (m == 0xC4) {
            int q = 0;
            while (q + 17 <= n) {
                const int tc = s[q] >> 4, th = s[q] & 15;
                if (tc > 1 || th > 3) return VBS_EINVAL;
                Huff& t = tc ? f->ac[th] : f->dc[th];
                int cnt = 0;
                t.bits[0] = 0;
                for (int i = 1; i <= 16; ++i) { t.bits[i] = s[q + i]; cnt += t.bits[i]; }
                if (cnt > 256 || q + 17 + cnt > n) return VBS_EINVAL;
                memcpy(t.vals, s + q + 17, cnt);
                t.build();
                if (!t.ok) return VBS_EINVAL;
                q += 17 + cnt;
            }
        } else if (m == 0xDB) {
            int q = 0;
            while (q < n) {
                const int pq = s[q] >> 4, tq = s[q] & 15;
                if (tq > 3 || pq > 1 || q + 1 + 64 * (pq + 1) > n) return VBS_EINVAL;
                for (int i = 0; i < 64; ++i)             // stored in zigzag order -> natural order
                    f->qt[tq][ZIGZAG[i]] = pq ? (uint16_t)be16(s + q + 1 + 2 * i) : s[q + 1 + i];
                f->qt_ok[tq] = true;
                q += 1 + 64 * (pq + 1);
            }
        } else if (m == 0xDD) {
            if (n < 2) return VBS_EINVAL;
            f->restart = be16(s);
        } else if (m == 0xDA) {
            if (n < 1 || s[0] != f->ncomp || n < 1 + 2 * f->ncomp + 3) return VBS_EINVAL;
            for (int c = 0; c < f->ncomp; ++c) {
                f->td[c] = s[2 + 2 * c] >> 4; f->ta[c] = s[2 + 2 * c] & 15;
                if (f->td[c] > 3 || f->ta[c] > 3 || s[1 + 2 * c] != f->id[c]) return VBS_EINVAL;   // (one interleaved scan, frame order)
            }
            const uint8_t* t = s + 1 + 2 * f->ncomp;
            if (t[0] != 0 || t[1] != 63 || t[2] != 0) return VBS_EINVAL;
            f->scan = d + p + len;
            f->end = d + size;
            break;
        }
        p += len;
    }
    if (!f->scan || !f->w) return VBS_EINVAL;
    for (int c = 0; c < f->ncomp; ++c) {
        if (!f->qt_ok[f->tq[c]]) return VBS_EINVAL;
        // a table the frame did not define: the standard one of that slot (0 luminance, 1 chrominance)
        if (!f->dc[f->td[c]].ok) { if (f->td[c] > 1) return VBS_EINVAL; f->dc[f->td[c]] = std_tables().dc[f->td[c]]; }
        if (!f->ac[f->ta[c]].ok) { if (f->ta[c] > 1) return VBS_EINVAL; f->ac[f->ta[c]] = std_tables().ac[f->ta[c]]; }
    }
    if (f->ncomp == 1) { f->hs[0] = f->vs[0] = 1; return VBS_OK; }
    // chroma 1x1, luma 1x1 / 2x1 / 2x2 (4:4:4, 4:2:2, 4:2:0)
    if (f->hs[1] != 1 || f->vs[1] != 1 || f->hs[2] != 1 || f->vs[2] != 1) return VBS_EINVAL;
    if (!((f->hs[0] == 1 && f->vs[0] == 1) || (f->hs[0] == 2 && f->vs[0] == 1) || (f->hs[0] == 2 && f->vs[0] == 2))) return VBS_EINVAL;
    return VBS_OK;
}

struct Bits {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc = 0;
    int cnt = 0;
    bool marker = false;                                 // a marker was met: only zeros follow
    inline void fill() {
        if (!marker && end - p >= 8) {                   // eight bytes at once when none of them is 0xFF (stuffing, markers)
            uint64_t v;
            memcpy(&v, p, 8);
            v = __builtin_bswap64(v);
            const uint64_t x = ~v;
            if (!((x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull)) {
                const int nb = (64 - cnt) >> 3;          // whole bytes that fit (cnt <= 56: at least one)
                if (nb > 0) {
                    const uint64_t top = nb == 8 ? v : (v >> (64 - 8 * nb)) << (64 - 8 * nb);
                    acc |= cnt ? top >> cnt : top;
                    cnt += 8 * nb;
                    p += nb;
                }
                return;
            }
        }
        while (cnt <= 56) {
            int b = 0;
            if (!marker && p < end) {
                b = *p;
                if (b == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) p += 2;
                    else { marker = true; b = 0; }
                } else ++p;
            }
            acc |= (uint64_t)b << (56 - cnt);
            cnt += 8;
        }
    }
    inline int peek(int n) { return (int)(acc >> (64 - n)); }
    inline void skip(int n) { acc <<= n; cnt -= n; }
    inline int get(int n) { if (!n) return 0; const int v = peek(n); skip(n); return v; }
    void reset() { acc = 0; cnt = 0; marker = false; }
};

inline int decode_sym(Bits& b, const Huff& t) {
    if (b.cnt < 16) b.fill();
    const uint16_t e = t.look[b.peek(9)];
    if (e) { b.skip(e >> 8); return e & 255; }
    int code = b.peek(10), l = 10;
    while (l <= 16 && code > t.maxcode[l]) { ++l; code = b.peek(l); }
    if (l > 16) return -1;
    b.skip(l);
    return t.vals[t.valptr[l] + code - t.mincode[l]];
}

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

// One frame's scan -> its blocks in the COMPACT form the device half reads.  Per block (numbered component after component,
// row-major over the component's padded block grid) one word in tab: (first word of the block in `ent`, relative to the
// frame's first) << 7 | count.  count <= 32: that many entry words (natural-order position << 16 | quantised value as uint16);
// count == 127: the block is dense, 64 int16 in natural order = 32 words.  A block never takes more than 32 words, so a frame
// never takes more than half of info[6] words, whatever its content.
int entropy(const Frame& f, uint32_t* ent, uint32_t* tab, int64_t* used) {
    const int hmax = f.hs[0], vmax = f.vs[0];
    const int mcux = (f.w + 8 * hmax - 1) / (8 * hmax), mcuy = (f.h + 8 * vmax - 1) / (8 * vmax);
    int bw[3], bh[3];
    int64_t base[3], off = 0;
    for (int c = 0; c < f.ncomp; ++c) {
        bw[c] = mcux * f.hs[c]; bh[c] = mcuy * f.vs[c];
        base[c] = off;
        off += (int64_t)bw[c] * bh[c];
    }
    Bits b{f.scan, f.end};
    int pred[3] = {0, 0, 0};
    int todo = f.restart;
    uint32_t at_word = 0;
    uint32_t e[64];
    for (int my = 0; my < mcuy; ++my)
        for (int mx = 0; mx < mcux; ++mx) {
            if (f.restart && todo == 0) {                // expect RSTn: byte-align, skip the marker, reset
                const uint8_t* q = b.p;
                // the reader may have run into the marker already (marker flag): find it from the current byte position
                while (q + 1 < f.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
                if (q + 1 >= f.end) return VBS_EINVAL;
                b.p = q + 2;
                b.reset();
                pred[0] = pred[1] = pred[2] = 0;
                todo = f.restart;
            }
            for (int c = 0; c < f.ncomp; ++c)
                for (int v = 0; v < f.vs[c]; ++v)
                    for (int hh = 0; hh < f.hs[c]; ++hh) {
                        const int64_t blk = base[c] + (int64_t)(my * f.vs[c] + v) * bw[c] + mx * f.hs[c] + hh;
                        uint32_t* dst = ent + at_word;   // (entries straight to their place: at most 64, 32 words are left)
                        int cnt = 0;
                        int s = decode_sym(b, f.dc[f.td[c]]);
                        if (s < 0 || s > 11) return VBS_EINVAL;
                        if (b.cnt < 16) b.fill();
                        const int diff = s ? extend(b.get(s), s) : 0;
                        pred[c] += diff;
                        if (pred[c]) e[cnt++] = (uint32_t)(uint16_t)(int16_t)pred[c];
                        const Huff& at = f.ac[f.ta[c]];
                        for (int k = 1; k < 64;) {
                            if (b.cnt < 32) b.fill();    // a code (<= 16 bits) and its value bits (<= 15)
                            const int32_t fa = at.fast[b.peek(9)];
                            if (fa) {                    // code + value within the nine bits
                                k += (fa >> 8) & 15;
                                if (k > 63) return VBS_EINVAL;
                                b.skip(fa & 255);
                                e[cnt++] = (uint32_t)ZIGZAG[k] << 16 | (uint32_t)(uint16_t)(int16_t)(fa >> 16);
                                ++k;
                                continue;
                            }
                            const int rs = decode_sym(b, at);
                            if (rs < 0) return VBS_EINVAL;
                            const int r = rs >> 4, sz = rs & 15;
                            if (!sz) {
                                if (r == 15) { k += 16; continue; }
                                break;                   // EOB
                            }
                            k += r;
                            if (k > 63) return VBS_EINVAL;
                            const int val = extend(b.get(sz), sz);
                            if (val) e[cnt++] = (uint32_t)ZIGZAG[k] << 16 | (uint32_t)(uint16_t)(int16_t)val;
                            ++k;
                        }
                        if (cnt <= 32) {
                            for (int i = 0; i < cnt; ++i) dst[i] = e[i];
                            tab[blk] = at_word << 7 | (uint32_t)cnt;
                            at_word += cnt;
                        } else {
                            int16_t d[64];
                            memset(d, 0, sizeof d);
                            for (int i = 0; i < cnt; ++i) d[e[i] >> 16] = (int16_t)(uint16_t)(e[i] & 0xFFFF);
                            memcpy(dst, d, sizeof d);
                            tab[blk] = at_word << 7 | 127u;
                            at_word += 32;
                        }
                    }
            if (f.restart) --todo;
        }
    *used = at_word;
    return VBS_OK;
}

// ---- device side ------------------------------------------------------------------------------------------------------
// jidctint.c (jpeg_idct_islow): CONST_BITS 13, PASS1_BITS 2; dequantisation folded in; output = range_limit(x + 128)
#define FIX_0_298631336 2446
#define FIX_0_390180644 3196
#define FIX_0_541196100 4433
#define FIX_0_765366865 6270
#define FIX_0_899976223 7373
#define FIX_1_175875602 9633
#define FIX_1_501321110 12299
#define FIX_1_847759065 15137
#define FIX_1_961570560 16069
#define FIX_2_053119869 16819
#define FIX_2_562915447 20995
#define FIX_3_072711026 25172

// (64-bit intermediates like libjpeg's JLONG: extreme coefficient blocks would wrap 32-bit products)
__device__ __forceinline__ void idct8(const int (&in)[8], int (&out)[8], int shift) {
    typedef long long L;
    // even part
    L z2 = in[2], z3 = in[6];
    L z1 = (z2 + z3) * FIX_0_541196100;
    L tmp2 = z1 + z3 * (-FIX_1_847759065);
    L tmp3 = z1 + z2 * FIX_0_765366865;
    z2 = in[0]; z3 = in[4];
    L tmp0 = (z2 + z3) * 8192;
    L tmp1 = (z2 - z3) * 8192;
    const L tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    // odd part
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    L z4 = tmp1 + tmp3;
    const L z5 = (z3 + z4) * FIX_1_175875602;
    tmp0 *= FIX_0_298631336; tmp1 *= FIX_2_053119869; tmp2 *= FIX_3_072711026; tmp3 *= FIX_1_501321110;
    z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447; z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    const L rnd = (L)1 << (shift - 1);
    out[0] = (int)((tmp10 + tmp3 + rnd) >> shift); out[7] = (int)((tmp10 - tmp3 + rnd) >> shift);
    out[1] = (int)((tmp11 + tmp2 + rnd) >> shift); out[6] = (int)((tmp11 - tmp2 + rnd) >> shift);
    out[2] = (int)((tmp12 + tmp1 + rnd) >> shift); out[5] = (int)((tmp12 - tmp1 + rnd) >> shift);
    out[3] = (int)((tmp13 + tmp0 + rnd) >> shift); out[4] = (int)((tmp13 - tmp0 + rnd) >> shift);
}

// One workgroup = 32 blocks of one component of one frame: 256 threads, thread = (block, column) in pass 1, (block, row) in
// pass 2.  The block's words (see `entropy`) are scattered, de-quantised, into LDS first.  plane: [n][ph][pw] u8 (whole blocks).
__global__ __launch_bounds__(256) void k_jpeg_idct(const uint32_t* __restrict__ ent, const uint32_t* __restrict__ tab,
                                                  const int64_t* __restrict__ frame_base, const unsigned short* __restrict__ qt,
                                                  unsigned char* __restrict__ plane, int tab_frame, int tab_base, int qt_index,
                                                  int nblocks, int bw, int64_t plane_frame, int pw) {
    __shared__ int cf[32][65];
    __shared__ int ws[32][8][9];
    const int n = blockIdx.y;
    const int lb = threadIdx.x >> 3, k = threadIdx.x & 7;
    const int b = blockIdx.x * 32 + lb;
    const unsigned short* q = qt + (int64_t)n * (3 * 64) + qt_index * 64;
    const bool live = b < nblocks;
    uint32_t word = 0;
    if (live) word = tab[(int64_t)n * tab_frame + tab_base + b];
    const uint32_t* src = ent + frame_base[n] + (word >> 7);
    const int cnt = (int)(word & 127u);
    if (live && cnt != 127) {
#pragma unroll
        for (int r = 0; r < 8; ++r) cf[lb][8 * r + k] = 0;
    }
    __syncthreads();
    if (live) {
        if (cnt == 127) {
            const short* d = reinterpret_cast<const short*>(src);
#pragma unroll
            for (int r = 0; r < 8; ++r) cf[lb][8 * r + k] = (int)d[8 * r + k] * (int)q[8 * r + k];
        } else {
            for (int j = k; j < cnt; j += 8) {
                const uint32_t e = src[j];
                const int pos = (int)(e >> 16) & 63;
                cf[lb][pos] = (int)(short)(e & 0xFFFFu) * (int)q[pos];
            }
        }
    }
    __syncthreads();
    if (live) {
        int in[8], out[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) in[r] = cf[lb][8 * r + k];
        idct8(in, out, 13 - 2);                          // columns: CONST_BITS - PASS1_BITS
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[lb][r][k] = out[r];
    }
    __syncthreads();
    if (live) {
        int in[8], out[8];
#pragma unroll
        for (int cidx = 0; cidx < 8; ++cidx) in[cidx] = ws[lb][k][cidx];
        idct8(in, out, 13 + 2 + 3);                      // rows: CONST_BITS + PASS1_BITS + 3
        const int by = b / bw, bx = b - by * bw;
        unsigned char* o = plane + (int64_t)n * plane_frame + (int64_t)(8 * by + k) * pw + 8 * bx;
        unsigned int lo = 0, hi = 0;
#pragma unroll
        for (int cidx = 0; cidx < 8; ++cidx) {
            const int v = min(max(out[cidx] + 128, 0), 255);
            if (cidx < 4) lo |= (unsigned)v << (8 * cidx); else hi |= (unsigned)v << (8 * (cidx - 4));
        }
        *reinterpret_cast<uint2*>(o) = make_uint2(lo, hi);
    }
}

// jdsample.c fancy upsampling + jdcolor.c.  mode 0: 4:4:4, 1: 4:2:2 (h2v1), 2: 4:2:0 (h2v2), 3: grayscale.
__device__ __forceinline__ int h2v1(const unsigned char* row, int cw, int x) {
    const int i = x >> 1, v = row[i];
    if (x & 1) return i + 1 < cw ? (3 * v + row[i + 1] + 2) >> 2 : v;
    return i > 0 ? (3 * v + row[i - 1] + 1) >> 2 : v;
}
__device__ __forceinline__ int h2v2(const unsigned char* near_, const unsigned char* far_, int cw, int x) {
    const int i = x >> 1;
    const int cur = 3 * near_[i] + far_[i];
    if (x & 1) {
        if (i + 1 < cw) return (3 * cur + 3 * near_[i + 1] + far_[i + 1] + 7) >> 4;
        return (4 * cur + 7) >> 4;
    }
    if (i > 0) return (3 * cur + 3 * near_[i - 1] + far_[i - 1] + 8) >> 4;
    return (4 * cur + 8) >> 4;
}

__global__ __launch_bounds__(256) void k_jpeg_color(const unsigned char* __restrict__ py, const unsigned char* __restrict__ pcb,
                                                   const unsigned char* __restrict__ pcr, int64_t yframe, int ypw, int64_t cframe,
                                                   int cpw, int cw, int ch, int mode, unsigned char* __restrict__ out,
                                                   int64_t out_frame, int64_t out_row, int W, int H) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), n = blockIdx.z;
    if (x >= W || y >= H) return;
    const int Y = py[(int64_t)n * yframe + (int64_t)y * ypw + x];
    unsigned char* o = out + (int64_t)n * out_frame + (int64_t)y * out_row + 3 * x;
    if (mode == 3) { o[0] = o[1] = o[2] = (unsigned char)Y; return; }
    const unsigned char* cb = pcb + (int64_t)n * cframe;
    const unsigned char* cr = pcr + (int64_t)n * cframe;
    int Cb, Cr;
    if (mode == 0) {
        Cb = cb[(int64_t)y * cpw + x]; Cr = cr[(int64_t)y * cpw + x];
    } else if (cw <= 2) {
        // jdsample.c takes the "fancy" upsamplers only for components more than two samples wide; narrower ones are replicated
        const int64_t at = (int64_t)(mode == 2 ? y >> 1 : y) * cpw + (x >> 1);
        Cb = cb[at]; Cr = cr[at];
    } else if (mode == 1) {
        Cb = h2v1(cb + (int64_t)y * cpw, cw, x); Cr = h2v1(cr + (int64_t)y * cpw, cw, x);
    } else {
        const int r = y >> 1;
        const int rf = (y & 1) ? min(r + 1, ch - 1) : max(r - 1, 0);     // the farther row: below for odd, above for even rows
        Cb = h2v2(cb + (int64_t)r * cpw, cb + (int64_t)rf * cpw, cw, x);
        Cr = h2v2(cr + (int64_t)r * cpw, cr + (int64_t)rf * cpw, cw, x);
    }
    const int cbx = Cb - 128, crx = Cr - 128;
    const int R = Y + ((91881 * crx + 32768) >> 16);                                   // FIX(1.40200)
    const int G = Y + ((-22554 * cbx - 46802 * crx + 32768) >> 16);                    // FIX(0.34414), FIX(0.71414)
    const int B = Y + ((116130 * cbx + 32768) >> 16);                                  // FIX(1.77200)
    o[0] = (unsigned char)min(max(B, 0), 255);
    o[1] = (unsigned char)min(max(G, 0), 255);
    o[2] = (unsigned char)min(max(R, 0), 255);
}

}  // namespace

// info[8]: width, height, components, luma h, luma v, restart interval, coefficients per frame, planes bytes per frame
extern "C" int vbs_mjpeg_probe(const uint8_t* jpeg, int64_t size, int32_t* info) {
    if (!jpeg || !info) return VBS_EINVAL;
    Frame f;
    const int rc = parse(jpeg, size, &f);
    if (rc != VBS_OK) return rc;
    const int mcux = (f.w + 8 * f.hs[0] - 1) / (8 * f.hs[0]), mcuy = (f.h + 8 * f.vs[0] - 1) / (8 * f.vs[0]);
    int64_t coefs = 0, planes = 0;
    for (int c = 0; c < f.ncomp; ++c) {
        const int64_t blocks = (int64_t)mcux * f.hs[c] * mcuy * f.vs[c];
        coefs += blocks * 64;
        planes += blocks * 64;
    }
    if (coefs / 2 >= ((int64_t)1 << 25)) return VBS_EINVAL;             // (a block's first word has 25 bits in its table word)
    info[0] = f.w; info[1] = f.h; info[2] = f.ncomp; info[3] = f.hs[0]; info[4] = f.vs[0]; info[5] = f.restart;
    info[6] = (int32_t)coefs; info[7] = (int32_t)planes;
    return VBS_OK;
}

// Entropy-decodes n frames (buf + offs[i], sizes[i]) on `threads` C++ threads.  Every frame must have the geometry of `info`
// (vbs_mjpeg_probe of the first).  ent: n * info[6] / 2 words; thread t packs its frames one after the other from word
// (first frame of t) * info[6] / 2 on, and reports (first word, words used) in regions[2 t], regions[2 t + 1] - only those
// spans need to reach the device.  tab [n][info[6] / 64], frame_base [n] (word index of a frame's first word in ent),
// qt [n][3][64] (natural order), status[i] = VBS_OK or VBS_EINVAL.  Returns the number of frames that failed.
extern "C" int vbs_mjpeg_entropy_batch(const uint8_t* buf, const int64_t* offs, const int32_t* sizes, int n, const int32_t* info,
                                       uint32_t* ent, uint32_t* tab, int64_t* frame_base, int64_t* regions, uint16_t* qt,
                                       int32_t* status, int threads) {
    if (!buf || !offs || !sizes || !info || !ent || !tab || !frame_base || !regions || !qt || !status || n < 0 || threads < 1)
        return VBS_EINVAL;
    const int64_t cap = info[6] / 2, nblk = info[6] / 64;
    for (int t = 0; t < threads; ++t) regions[2 * t] = regions[2 * t + 1] = 0;
    auto work = [&](int t, int t0, int t1) {
        int64_t at = (int64_t)t0 * cap;
        regions[2 * t] = at;
        for (int i = t0; i < t1; ++i) {
            Frame f;
            int rc = parse(buf + offs[i], sizes[i], &f);
            if (rc == VBS_OK && (f.w != info[0] || f.h != info[1] || f.ncomp != info[2] || f.hs[0] != info[3] || f.vs[0] != info[4]))
                rc = VBS_EINVAL;
            int64_t used = 0;
            frame_base[i] = at;
            if (rc == VBS_OK) rc = entropy(f, ent + at, tab + (int64_t)i * nblk, &used);
            if (rc == VBS_OK) {
                at += used;
                for (int c = 0; c < 3; ++c) memcpy(qt + ((int64_t)i * 3 + c) * 64, f.qt[f.tq[c < f.ncomp ? c : 0]], 128);
            }
            status[i] = rc;
        }
        regions[2 * t + 1] = at - regions[2 * t];
    };
    const int nt = std::max(1, std::min(threads, n));
    if (nt <= 1) work(0, 0, n);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(work, t, (int)((int64_t)n * t / nt), (int)((int64_t)n * (t + 1) / nt));
        for (auto& x : th) x.join();
    }
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += status[i] != VBS_OK;
    return bad;
}

// Device half: ent / tab / frame_base / qt as vbs_mjpeg_entropy_batch left them (DEVICE copies) -> BGR frames out [n] (out_frame
// / out_row strides in bytes, 3 bytes per pixel); planes = scratch of n * info[7] bytes.  Asynchronous on `stream`.
extern "C" int vbs_mjpeg_reconstruct(const uint32_t* ent, const uint32_t* tab, const int64_t* frame_base, const uint16_t* qt, int n,
                                     const int32_t* info, uint8_t* planes, uint8_t* out, int64_t out_frame, int64_t out_row,
                                     void* stream) {
    if (!ent || !tab || !frame_base || !qt || !info || !planes || !out || n < 0 || n > 65535) return VBS_EINVAL;   // (frames = a grid dimension)
    if (n == 0) return VBS_OK;
    hipStream_t s = (hipStream_t)stream;
    const int W = info[0], H = info[1], nc = info[2], hs = info[3], vs = info[4];
    const int mcux = (W + 8 * hs - 1) / (8 * hs), mcuy = (H + 8 * vs - 1) / (8 * vs);
    const int ybw = mcux * hs, ybh = mcuy * vs, cbw = mcux, cbh = mcuy;
    const int64_t yblocks = (int64_t)ybw * ybh, cblocks = (int64_t)cbw * cbh;
    const int64_t plane_frame = info[7];
    const int tab_frame = info[6] / 64;
    // planes of one frame: Y [8 ybh][8 ybw], then Cb, Cr [8 cbh][8 cbw]
    hipLaunchKernelGGL(k_jpeg_idct, dim3((unsigned)((yblocks + 31) / 32), n), dim3(256), 0, s, ent, tab, frame_base, qt, planes,
                       tab_frame, 0, 0, (int)yblocks, ybw, plane_frame, 8 * ybw);
    if (nc == 3) {
        hipLaunchKernelGGL(k_jpeg_idct, dim3((unsigned)((cblocks + 31) / 32), n), dim3(256), 0, s, ent, tab, frame_base, qt,
                           planes + yblocks * 64, tab_frame, (int)yblocks, 1, (int)cblocks, cbw, plane_frame, 8 * cbw);
        hipLaunchKernelGGL(k_jpeg_idct, dim3((unsigned)((cblocks + 31) / 32), n), dim3(256), 0, s, ent, tab, frame_base, qt,
                           planes + yblocks * 64 + cblocks * 64, tab_frame, (int)(yblocks + cblocks), 2, (int)cblocks, cbw,
                           plane_frame, 8 * cbw);
    }
    const int mode = nc == 1 ? 3 : (hs == 1 ? 0 : (vs == 1 ? 1 : 2));
    const int cw = (W + hs - 1) / hs, ch = (H + vs - 1) / vs;        // chroma samples that belong to the image
    hipLaunchKernelGGL(k_jpeg_color, dim3((W + 63) / 64, (H + 3) / 4, n), dim3(256), 0, s, planes, planes + yblocks * 64,
                       planes + yblocks * 64 + cblocks * 64, plane_frame, 8 * ybw, plane_frame, 8 * cbw, cw, ch, mode, out, out_frame,
                       out_row, W, H);
    return hipGetLastError() == hipSuccess ? VBS_OK : VBS_EHIP;
}
