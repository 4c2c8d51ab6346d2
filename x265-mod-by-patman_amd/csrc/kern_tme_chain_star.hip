// kern_tme_chain_star.hip -- tme_chain.inc with the STAR search code
// r04 (profiles/r04_chain_lds_ab.txt): the MVD cost slice every lane group keeps in LDS is +-256 quarter-pels here (+-512 in the batch kernels; what lies beyond is read from
// memory, me_body.inc cost1): 32 groups x 2 KB made a workgroup of the 8-lane kernels 80 KB -- two per CU whatever the registers allowed; at 52 KB three fit.
#ifndef XH_COST_R
#define XH_COST_R 256
#endif
#define XH_ME_WIDE 1
#define XH_CHAIN_STARK 1
#define XH_CHAIN_ENTRY xh_tme_chain_star
// (built for 4 workgroups per CU = 128 registers, tme_chain.inc: the kernels spill 336-400 bytes per lane then, and are still faster than with 256 registers and no spills --
//  1080p preset slow 5.3 against 6.4 ms per picture, slower 8.1 against 9.4 -- unlike the short HEX / UMH chains)
#include "tme_chain.inc"
