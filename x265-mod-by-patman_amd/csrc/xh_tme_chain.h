// xh_tme_chain.h -- the interface between x265hip_tme_frame (kern_tme.hip) and the chain kernels (tme_chain.inc): one launch runs, for every CTU of the picture, the whole
// chain of every PU shape that shares a kernel configuration.
#pragma once
#include "xh_tme_glue.h"

struct xh_chain_ref
{
    const pixel* mePlane; const pixel* mePhase; const pixel* reconPhase;        // the searched plane (weighted when the reference is), its phase planes, the reconstruction's
    const x265hip_inter_choice* refTable; const int16_t* lowresMv;
};
struct xh_chain_args
{
    xh::tme::Slice s;
    xh::tme::Lambdas lambdas;
    const x265hip_tme_step* steps; int nSteps, nCtu;      // the schedule (device copy); the CTUs of this launch: ctuFirst .. ctuFirst + nCtu - 1 of the picture
    int ctuFirst;
    int keys[8];                                           // cuSize * 8 + part of the shapes of this launch (blockIdx.y)
    int keyRow[8], nLevels[8], firstStep[8];               // per shape: its row of `sched`, its number of levels, one of its entries (the shape's PU sizes)
    const int16_t* sched;                                  // [shape row][XH_CHAIN_LEVELS][XH_CHAIN_WIDTH]: the entries of a shape by level (-1: none) -- entries of one level
                                                           // have no neighbour among each other or in later levels that precedes them in the schedule: they run side by side
    const uint8_t* later;                                  // per entry: bit d = neighbour d comes later in the schedule (its record is read from tableInit)
    const x265hip_inter_choice* tableInit;                 // the table as it was when the picture started
    const pixel* cur; int64_t planeElems;
    xh_chain_ref refs[2][X265HIP_MAX_REF];
    x265hip_inter_choice* table; const int16_t* areaBest; const x265hip_tme_temporal* temporal; const uint8_t* qpIndex;
    const uint16_t* costRows; int costHalf; const float* bitsCentre; int bitsHalf;
    int searchRange, method, subme;
    int dbg;                                               // me_one's debug exits; 0
};

constexpr int XH_CHAIN_LEVELS = 64, XH_CHAIN_WIDTH = 4;
// entries of a shape that a CTU runs side by side (lane groups per CTU) in each kernel configuration
inline int xh_chain_width(int config) { return config <= 1 ? 4 : config <= 4 ? 2 : 1; }

// kernel configuration of a shape (the lanes per PU and LDS per PU dispatch_me uses for its PUs; both partitions of an AMP shape run in the larger one's)
inline int xh_chain_config(int cuSize, int part)
{
    if (cuSize == 8) return part == 0 ? 0 : 1;
    if (cuSize == 16) return part == 0 ? 2 : part <= 2 ? 3 : 4;
    if (cuSize == 32) return part == 0 ? 5 : 6;
    return 7;
}
int xh_tme_chain_hex(void* stream, int config, const xh_chain_args* args, int nKeys, bool packed);     // DIA / HEX / FULL
int xh_tme_chain_star(void* stream, int config, const xh_chain_args* args, int nKeys, bool packed);    // STAR
int xh_tme_chain_umh(void* stream, int config, const xh_chain_args* args, int nKeys, bool packed);     // UMH
