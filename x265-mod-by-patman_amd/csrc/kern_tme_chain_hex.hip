// kern_tme_chain_hex.hip -- tme_chain.inc with the DIA / HEX / FULL search code
#define XH_ME_WIDE 1
#ifndef XH_CHAIN_MINWG
#define XH_CHAIN_MINWG 2                  // the short HEX searches lose more to spills than they gain from a fourth wavefront per SIMD (preset medium 720p: 1.9 vs 2.2 ms per picture)
#endif
#define XH_CHAIN_STARK 0
#define XH_CHAIN_ENTRY xh_tme_chain_hex
#include "tme_chain.inc"
