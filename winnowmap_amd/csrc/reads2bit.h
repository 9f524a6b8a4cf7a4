// reads2bit.h — the read codes of a mini-batch as they live in HBM (and cross PCIe): 2 bits per base + 1 ambiguity bit per base = 0.375 B / base
// instead of one byte. Base p — an index into the mini-batch's concatenated reads, slab offset included — has
//     code  = (pk[p >> 5] >> 2 * (p & 31)) & 3        (A C G T = 0 1 2 3, the reference's 2-bit k-mer digits, src/sketch.c:164-165)
//     N     = (nm[p >> 6] >> (p & 63)) & 1            (code 4 of seq_nt4_table, src/sketch.c:19-36; its two code bits are 0)
// so that the k bases ending at a position are ONE shifted 64-bit window of pk (k <= 28: 56 bits) — the forward k-mer is the window with its 2-bit
// groups reversed, the reverse complement k-mer its complement — where the byte layout needed k loads.
// Host side: wm_pack_codes (8 bases per step, no BMI2 needed). Device side: rd2_code / rd2_window / rd2_rev (simt.h types; also compiled on the
// wavefront emulator).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>

// a sequence offset with this bit set (wm_sketch_job_t::seq_off) is a BASE index into the resident packed reads, not a byte offset into the call's staged bytes
#define WM_RD_PACKED_BIT ((uint64_t)1 << 62)

// words of pk / nm that hold n bases (+ one word of slack each: rd2_window reads the word after the last base's)
static inline size_t wm_pk_words(size_t n) { return (n + 31) / 32 + 1; }
static inline size_t wm_nm_words(size_t n) { return (n + 63) / 64 + 1; }

// codes[0 .. n) (values 0..4; anything >= 4 counts as ambiguous) -> the 64-base blocks of pk / nm from word 0 on: 2 * ceil(n / 64) words of pk, ceil(n / 64) of
// nm, tail bits zero. A range that starts at a multiple of 64 bases can be packed on its own (threads pack disjoint ranges of one buffer).
static inline void wm_pack_blocks(const uint8_t *codes, size_t n, uint64_t *pk, uint64_t *nm)
{
	const size_t nw = (n + 63) / 64;                                        // 64-base blocks, the last one possibly short
	for (size_t b = 0; b < nw; ++b) {
		const size_t at = b * 64, len = n - at < 64 ? n - at : 64;
		uint8_t buf[64];
		const uint8_t *src = codes + at;
		if (len < 64) { memset(buf, 0, sizeof(buf)); memcpy(buf, src, len); src = buf; }
		uint64_t p0 = 0, p1 = 0, m = 0;
		for (int g = 0; g < 8; ++g) {
			uint64_t x;
			memcpy(&x, src + 8 * g, 8);
			const uint64_t hi6 = x & 0xfcfcfcfcfcfcfcfcULL;                  // any of bits 2..7 set: the code is >= 4 (wm_reads_upload accepts arbitrary bytes)
			const uint64_t amb = ((hi6 | hi6 >> 1 | hi6 >> 2 | hi6 >> 3 | hi6 >> 4 | hi6 >> 5) >> 2) & 0x0101010101010101ULL;   // bit 0 of byte i: base i is ambiguous
			m |= ((amb * 0x0102040810204080ULL) >> 56) << (8 * g);          // the eight flags gathered into one byte, base 8g + i -> bit 8g + i
			x &= 0x0303030303030303ULL & ~(amb * 3);                         // an ambiguous base packs as 0
			const uint64_t y = x | x >> 6 | x >> 12 | x >> 18;               // bytes 0 and 4 of y: bases 0-3 and 4-7, two bits each
			const uint64_t h = (y & 0xff) | ((y >> 24) & 0xff00);
			if (g < 4) p0 |= h << (16 * g); else p1 |= h << (16 * (g - 4));
		}
		pk[2 * b] = p0; pk[2 * b + 1] = p1; nm[b] = m;
	}
}
// the whole of codes[0 .. n): wm_pk_words(n) / wm_nm_words(n) words, slack included
static inline void wm_pack_codes(const uint8_t *codes, size_t n, uint64_t *pk, uint64_t *nm)
{
	wm_pack_blocks(codes, n, pk, nm);
	for (size_t i = 2 * ((n + 63) / 64); i < wm_pk_words(n); ++i) pk[i] = 0;
	for (size_t i = (n + 63) / 64; i < wm_nm_words(n); ++i) nm[i] = 0;
}

// the 0..4 code of base p (host side: tests, debugging)
static inline int wm_rd_code(const uint64_t *pk, const uint64_t *nm, uint64_t p)
{
	return (nm[p >> 6] >> (p & 63)) & 1 ? 4 : (int)((pk[p >> 5] >> (2 * (p & 31))) & 3);
}

#ifdef WM_DEV
namespace wmk {
using namespace simt;
// the 0..4 code of base p
WM_DEV V<int> rd2_code(const uint64_t *pk, const uint64_t *nm, V<long long> p)
{
	const V<uint64_t> w = gld(pk, p >> 5), m = gld(nm, p >> 6);
	const V<int> c = cast<int>((w >> cast<int>((p & 31LL) << 1)) & (uint64_t)3);
	const V<int> amb = cast<int>((m >> cast<int>(p & 63LL)) & (uint64_t)1);
	return sel(amb != 0, V<int>(4), c);
}
WM_DEV vbool rd2_is_n(const uint64_t *nm, V<long long> p) { return cast<int>((gld(nm, p >> 6) >> cast<int>(p & 63LL)) & (uint64_t)1) != 0; }
// the 32 bases from s on: base s + m in bits 2m
WM_DEV V<uint64_t> rd2_window(const uint64_t *pk, V<long long> s)
{
	const V<long long> wi = s >> 5;
	const V<int> sh = cast<int>((s & 31LL) << 1);
	const V<uint64_t> w0 = gld(pk, wi), w1 = gld(pk, wi + 1LL);
	return (w0 >> sh) | ((w1 << 1) << (V<int>(63) - sh));                            // (two shifts: sh = 0 must contribute nothing of w1)
}
// 2-bit groups of a word in reverse order (group m -> group 31 - m)
WM_DEV V<uint64_t> rd2_rev(V<uint64_t> x)
{
	x = ((x >> 2) & (uint64_t)0x3333333333333333ULL) | ((x & (uint64_t)0x3333333333333333ULL) << 2);
	x = ((x >> 4) & (uint64_t)0x0f0f0f0f0f0f0f0fULL) | ((x & (uint64_t)0x0f0f0f0f0f0f0f0fULL) << 4);
	x = ((x >> 8) & (uint64_t)0x00ff00ff00ff00ffULL) | ((x & (uint64_t)0x00ff00ff00ff00ffULL) << 8);
	x = ((x >> 16) & (uint64_t)0x0000ffff0000ffffULL) | ((x & (uint64_t)0x0000ffff0000ffffULL) << 16);
	return (x >> 32) | (x << 32);
}
} // namespace wmk
#endif
