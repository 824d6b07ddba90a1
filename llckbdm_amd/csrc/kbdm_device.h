// Device-side descriptors shared by the kernels and the host API of libkbdm_hip.so.
#pragma once
#include "kb_complex.hpp"

namespace kb {

// Per-item m x m (or l x l) complex work buffers, column-major.  Roles over the pipeline:
//   A : U^{p-1} -> bidiagonal form + reflectors -> L (sorted left vectors) -> B = R_ Dsqi P
//   Q : Q (left accumulation) -> T1 = U^p R_ -> Qh (Hessenberg basis) -> T = U0 B
//   P : P (right accumulation) -> W (reduced matrix, then Hessenberg + reflectors) -> G = Dsqi Qh X
//   R : R (sorted right vectors)
//   H : Hessenberg work copy for the QR iteration -> X (eigenvectors in the Hessenberg basis)
enum { KB_BUF_A = 0, KB_BUF_Q, KB_BUF_P, KB_BUF_R, KB_BUF_H, KB_NBUF };

// Per-item vector arena (doubles), slot s starts at voff + s * vstride.
enum {
    KB_V_D = 0,      // bidiagonal diagonal
    KB_V_E = 1,      // bidiagonal superdiagonal
    KB_V_S = 2,      // sorted singular values
    KB_V_DSQI = 3,   // 1/sqrt(s) (or Tikhonov form)
    KB_V_TAUQ = 4,   // complex, 2 slots (reused for the Hessenberg taus)
    KB_V_TAUP = 6,   // complex, 2 slots
    KB_V_MISC = 8,   // [0] = ||H||_inf
    KB_V_SLOTS = 9
};

struct KbItem {
    int m, l, sig, pad0;
    long long off[KB_NBUF];   // element (complex) offsets into the matrix arena
    long long voff;           // double offset into the vector arena
    int vstride, pad;
    long long line_off;       // offset of this item's lines / mu / keep (units: lines)
    long long sv_off;         // offset of this item's singular values
    long long hk_off;         // offset (complex elements) into dense per-item m*m outputs of the stage APIs
    long long dc_off;         // offset (doubles) of this item's divide-and-conquer workspace (kb_bdsdc.hpp: DcWs)
    double q;
};

struct HankelOut {
    cd* base;
    int shift;          // U^{shift}[i,j] = c[i + j + shift]
    int buf;            // KB_BUF_* when use_item_off
    int use_item_off;   // 1: base + item.off[buf] (pipeline), 0: base + item.hk_off (stage API)
};

}  // namespace kb

// status bits (same values as KBDM_STAT_* in include/kbdm_hip.h)
#define KB_STAT_SVD_NOCONV 1
#define KB_STAT_EIG_NOCONV 2
#define KB_STAT_INVIT_WEAK 4
