// kern_tme_chain_umh.hip -- tme_chain.inc with the UMH search code
// r04 (profiles/r04_chain_lds_ab.txt): the MVD cost slice every lane group keeps in LDS is +-256 quarter-pels here (+-512 in the batch kernels; what lies beyond is read from
// memory, me_body.inc cost1): 32 groups x 2 KB made a workgroup of the 8-lane kernels 80 KB -- two per CU whatever the registers allowed; at 52 KB three fit.
#ifndef XH_COST_R
#define XH_COST_R 256
#endif
#ifndef XH_CHAIN_MINWG
#define XH_CHAIN_MINWG 3
#endif
#define XH_ME_WIDE 1
#define XH_CHAIN_STARK 2
#define XH_CHAIN_ENTRY xh_tme_chain_umh
#include "tme_chain.inc"
