// kern_tme_chain_star.hip -- tme_chain.inc with the STAR search code
#define XH_ME_WIDE 1
#define XH_CHAIN_STARK 1
#define XH_CHAIN_ENTRY xh_tme_chain_star
// (built for 4 workgroups per CU = 128 registers, tme_chain.inc: the kernels spill 336-400 bytes per lane then, and are still faster than with 256 registers and no spills --
//  1080p preset slow 5.3 against 6.4 ms per picture, slower 8.1 against 9.4 -- unlike the short HEX / UMH chains)
#include "tme_chain.inc"
