/* C restatement of Philox4x32-10 (Salmon et al., SC'11; Random123 philox4x32_R(10)).
 * TEST INFRASTRUCTURE (see oracle/__init__.py): an independent check of oracle/philox.py and of
 * the device generator in bayeslms_amd/csrc/blm_device.h.  Built by oracle/Makefile. */
#include <stdint.h>

void blm_oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* raw 32-bit words of blocks [0, nblk) of stream (seed, stream, step) */
void blm_oracle_philox_words(uint64_t seed, uint32_t stream, uint32_t step, uint64_t nblk, uint32_t* out) {
  for (uint64_t b = 0; b < nblk; ++b) {
    const uint32_t ctr[4] = {(uint32_t)b, (uint32_t)(b >> 32), stream, step};
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    blm_oracle_philox4x32_10(ctr, key, out + 4 * b);
  }
}
