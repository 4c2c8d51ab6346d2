// kern_tme_chain_hex.hip -- tme_chain.inc with the DIA / HEX / FULL search code
// r04 (profiles/r04_chain_lds_ab.txt): the MVD cost slice every lane group keeps in LDS is +-256 quarter-pels here (+-512 in the batch kernels; what lies beyond is read from
// memory, me_body.inc cost1): 32 groups x 2 KB made a workgroup of the 8-lane kernels 80 KB -- two per CU whatever the registers allowed; at 52 KB three fit.
#ifndef XH_COST_R
#define XH_COST_R 256
#endif
#define XH_ME_WIDE 1
#ifndef XH_CHAIN_MINWG
#define XH_CHAIN_MINWG 3                  // three workgroups per CU (168 registers, <= 16 spilled; r04: preset medium 1080p 1.8 -> 1.56 ms per picture once the LDS allows three); a fourth
                                          // (128 registers) loses more to spills than it gains for the short HEX searches (preset medium 720p: 1.9 vs 2.2 ms per picture)
#endif
#define XH_CHAIN_STARK 0
#define XH_CHAIN_ENTRY xh_tme_chain_hex
#include "tme_chain.inc"
