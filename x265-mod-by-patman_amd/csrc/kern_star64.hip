// kern_star64.hip -- STAR for 64x64 PUs with the PU's whole search window in LDS (merange <= 57): star64_body.inc with the band of the pattern search
#define XS_NJ 23
#define XS_NIC 24
#include "star64_body.inc"
