// kern_tme_chain_hex.hip -- tme_chain.inc with the DIA / HEX / FULL search code
#define XH_ME_WIDE 1
#define XH_CHAIN_STARK 0
#define XH_CHAIN_ENTRY xh_tme_chain_hex
#include "tme_chain.inc"
