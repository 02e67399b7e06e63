// Host-side helpers of the input path (no device code): CRC-32C for the TFRecord framing of the reference's CelebA
// files (vae/data.py:93-100 writes them with tf.io.TFRecordWriter, :123-131 reads them back; the record format --
// uint64 length | masked crc32c(length) | data | masked crc32c(data) -- is TensorFlow's public one).
#include <stdint.h>
#include <stddef.h>
#include "../../include/splitvae.h"

namespace {
struct Crc32cTables {
  uint32_t t[8][256];
  Crc32cTables() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;   // reflected Castagnoli polynomial
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
  }
};
const Crc32cTables& tables() {
  static const Crc32cTables T;
  return T;
}
}  // namespace

// CRC-32C (iSCSI / Castagnoli), slicing-by-8: check value 0xE3069283 for "123456789"
extern "C" uint32_t sv_crc32c(const void* data, int64_t n) {
  const Crc32cTables& T = tables();
  const uint8_t* p = (const uint8_t*)data;
  uint32_t c = 0xFFFFFFFFu;
  while (n > 0 && ((uintptr_t)p & 7)) { c = T.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8); --n; }
  while (n >= 8) {
    const uint64_t v = *(const uint64_t*)p ^ c;       // little-endian host
    c = T.t[7][v & 0xFF] ^ T.t[6][(v >> 8) & 0xFF] ^ T.t[5][(v >> 16) & 0xFF] ^ T.t[4][(v >> 24) & 0xFF] ^
        T.t[3][(v >> 32) & 0xFF] ^ T.t[2][(v >> 40) & 0xFF] ^ T.t[1][(v >> 48) & 0xFF] ^ T.t[0][(v >> 56) & 0xFF];
    p += 8; n -= 8;
  }
  while (n-- > 0) c = T.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

// the TFRecord mask: rotate right by 15, add 0xa282ead8
extern "C" uint32_t sv_masked_crc32c(const void* data, int64_t n) {
  const uint32_t c = sv_crc32c(data, n);
  return ((c >> 15) | (c << 17)) + 0xa282ead8u;
}
