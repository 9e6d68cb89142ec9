/* ORACLE - TEST INFRASTRUCTURE ONLY (see gl.h header).
 * Byte writer / reader of the proof wire format (plonky2 util::serialization::Buffer). */
#ifndef NLX_ORACLE_BYTES_H
#define NLX_ORACLE_BYTES_H
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include "gl.h"

typedef struct { uint8_t* p; size_t len, cap; int overflow; } wbuf;
static inline void w_bytes(wbuf* w, const void* src, size_t n) {
    if (w->len + n > w->cap) { w->overflow = 1; return; }
    if (n) memcpy(w->p + w->len, src, n);
    w->len += n;
}
static inline void w_u64s(wbuf* w, const uint64_t* v, size_t n) { w_bytes(w, v, n * 8); } /* little-endian host */
static inline void w_u8(wbuf* w, uint8_t v) { w_bytes(w, &v, 1); }
/* plonky2 util::serialization Write::write_usize: a usize travels as 8 little-endian bytes */
static inline void w_usize(wbuf* w, uint64_t v) { w_bytes(w, &v, 8); }
typedef struct { const uint8_t* p; size_t len, pos; int bad; } rbuf;
static inline void r_bytes(rbuf* r, void* dst, size_t n) {
    if (r->pos + n > r->len) { r->bad = 1; memset(dst, 0, n); return; }
    memcpy(dst, r->p + r->pos, n);
    r->pos += n;
}
static inline void r_u64s(rbuf* r, uint64_t* v, size_t n) {
    r_bytes(r, v, n * 8);
    for (size_t i = 0; i < n; i++) if (v[i] >= GL_P) r->bad = 1; /* read_field rejects non-canonical */
}

#endif
