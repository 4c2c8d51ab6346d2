// kern_tme_chain_star.hip -- tme_chain.inc with the STAR search code
#define XH_ME_WIDE 1
#define XH_CHAIN_STARK 1
#define XH_CHAIN_ENTRY xh_tme_chain_star
#include "tme_chain.inc"
