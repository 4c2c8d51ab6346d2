// kern_tme_chain_umh.hip -- tme_chain.inc with the UMH search code
#define XH_CHAIN_MINWG 2
#define XH_ME_WIDE 1
#define XH_CHAIN_STARK 2
#define XH_CHAIN_ENTRY xh_tme_chain_umh
#include "tme_chain.inc"
