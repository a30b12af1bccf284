// Side-chain build programs (AMBER ff14SB geometry) as __constant__ tables.
// Same facts as /root/reference/protein_transformer/protein/SidechainBuildInfo.py:1-574,
// flattened to "atom k of residue type r hangs off slots (pa,pb,pc) with bond, angle, torsion".
// Slots: 0 N, 1 CA, 2 C, 3 O, 4+k side-chain atom k.  The first atom (CB) has the implicit
// parents (C of previous residue, N, CA) or, for the first residue, (N of next residue, C, CA)
// (StructureBuilder.py:205-216).
#pragma once

#define PT_MAX_SC 10
#define PT_TORS_ABSENT 0
#define PT_TORS_PRED 1   // predicted chi, angle column 6+k
#define PT_TORS_INFER 2  // previous torsion - pi (planar partner)
#define PT_TORS_CONST 3

struct PtScAtom {
  float bond;
  float angle;
  float tors_const;
  signed char kind, pa, pb, pc;
};

#define PT_PI_F 3.141592653589793f
#define PT_CB {1.526f, 1.9146261894377796f, 0.f, PT_TORS_PRED, -1, -1, -1}
#define PT_T 1.911135530933791f
#define PT_R 2.0943951023931953f
#define PT_P(b, a, x, y, z) {b, a, 0.f, PT_TORS_PRED, x, y, z}
#define PT_I(b, a, x, y, z) {b, a, 0.f, PT_TORS_INFER, x, y, z}
#define PT_C(b, a, t, x, y, z) {b, a, t, PT_TORS_CONST, x, y, z}
#define PT_NONE {0.f, 0.f, 0.f, PT_TORS_ABSENT, 0, 0, 0}

__constant__ int c_pt_nsc[20] = {1, 2, 4, 5, 7, 0, 6, 4, 5, 4, 4, 4, 3, 5, 7, 2, 3, 3, 10, 8};
static const int h_pt_nsc[20] = {1, 2, 4, 5, 7, 0, 6, 4, 5, 4, 4, 4, 3, 5, 7, 2, 3, 3, 10, 8};

__constant__ PtScAtom c_pt_sc[20][PT_MAX_SC] = {
    /* A */ {PT_CB},
    /* C */ {PT_CB, PT_P(1.81f, 1.8954275676658419f, 0, 1, 4)},
    /* D */ {PT_CB, PT_P(1.522f, 1.9390607989657f, 0, 1, 4), PT_P(1.25f, 2.0420352248333655f, 1, 4, 5),
             PT_I(1.25f, 2.0420352248333655f, 1, 4, 5)},
    /* E */ {PT_CB, PT_P(1.526f, PT_T, 0, 1, 4), PT_P(1.522f, 1.9390607989657f, 1, 4, 5),
             PT_P(1.25f, 2.0420352248333655f, 4, 5, 6), PT_I(1.25f, 2.0420352248333655f, 4, 5, 6)},
    /* F */ {PT_CB, PT_P(1.51f, 1.9896753472735358f, 0, 1, 4), PT_P(1.4f, PT_R, 1, 4, 5),
             PT_C(1.4f, PT_R, PT_PI_F, 4, 5, 6), PT_C(1.4f, PT_R, 0.f, 5, 6, 7), PT_C(1.4f, PT_R, 0.f, 6, 7, 8),
             PT_C(1.4f, PT_R, 0.f, 7, 8, 9)},
    /* G */ {PT_NONE},
    /* H */ {PT_CB, PT_P(1.504f, 1.9739673840055867f, 0, 1, 4), PT_P(1.385f, PT_R, 1, 4, 5),
             PT_C(1.343f, 1.8849555921538759f, PT_PI_F, 4, 5, 6), PT_C(1.335f, 1.8849555921538759f, 0.f, 5, 6, 7),
             PT_C(1.394f, 1.8849555921538759f, 0.f, 6, 7, 8)},
    /* I */ {PT_CB, PT_P(1.526f, PT_T, 0, 1, 4), PT_P(1.526f, PT_T, 1, 4, 5), PT_P(1.526f, PT_T, 0, 1, 4)},
    /* K */ {PT_CB, PT_P(1.526f, PT_T, 0, 1, 4), PT_P(1.526f, PT_T, 1, 4, 5), PT_P(1.526f, PT_T, 4, 5, 6),
             PT_P(1.471f, 1.9408061282176945f, 5, 6, 7)},
    /* L */ {PT_CB, PT_P(1.526f, PT_T, 0, 1, 4), PT_P(1.526f, PT_T, 1, 4, 5), PT_P(1.526f, PT_T, 1, 4, 5)},
    /* M */ {PT_CB, PT_P(1.526f, PT_T, 0, 1, 4), PT_P(1.81f, 2.0018926520374962f, 1, 4, 5),
             PT_P(1.81f, 1.726130630222392f, 4, 5, 6)},
    /* N */ {PT_CB, PT_P(1.522f, 1.9390607989657f, 0, 1, 4), PT_P(1.229f, 2.101376419401173f, 1, 4, 5),
             PT_I(1.335f, 2.035053907825388f, 1, 4, 5)},
    /* P */ {PT_CB, PT_P(1.526f, PT_T, 0, 1, 4), PT_P(1.526f, PT_T, 1, 4, 5)},
    /* Q */ {PT_CB, PT_P(1.526f, PT_T, 0, 1, 4), PT_P(1.522f, 1.9390607989657f, 1, 4, 5),
             PT_P(1.229f, 2.101376419401173f, 4, 5, 6), PT_I(1.335f, 2.035053907825388f, 4, 5, 6)},
    /* R */ {PT_CB, PT_P(1.526f, PT_T, 0, 1, 4), PT_P(1.526f, PT_T, 1, 4, 5), PT_P(1.463f, 1.9408061282176945f, 4, 5, 6),
             PT_P(1.34f, 2.150245638457014f, 5, 6, 7), PT_P(1.34f, PT_R, 6, 7, 8), PT_I(1.34f, PT_R, 6, 7, 8)},
    /* S */ {PT_CB, PT_P(1.41f, PT_T, 0, 1, 4)},
    /* T */ {PT_CB, PT_P(1.41f, PT_T, 0, 1, 4), PT_P(1.526f, PT_T, 0, 1, 4)},
    /* V */ {PT_CB, PT_P(1.526f, PT_T, 0, 1, 4), PT_P(1.526f, PT_T, 0, 1, 4)},
    /* W */ {PT_CB, PT_P(1.495f, 2.0176006153054447f, 0, 1, 4), PT_P(1.352f, 2.181661564992912f, 1, 4, 5),
             PT_C(1.381f, 1.8971728969178363f, PT_PI_F, 4, 5, 6), PT_C(1.38f, 1.9477874452256716f, 0.f, 5, 6, 7),
             PT_C(1.4f, 2.3177972466484698f, PT_PI_F, 6, 7, 8), PT_C(1.4f, PT_R, PT_PI_F, 7, 8, 9),
             PT_C(1.4f, PT_R, 0.f, 8, 9, 10), PT_C(1.4f, PT_R, 0.f, 9, 10, 11), PT_C(1.404f, PT_R, 0.f, 10, 11, 12)},
    /* Y */ {PT_CB, PT_P(1.51f, 1.9896753472735358f, 0, 1, 4), PT_P(1.4f, PT_R, 1, 4, 5),
             PT_C(1.4f, PT_R, PT_PI_F, 4, 5, 6), PT_C(1.409f, PT_R, 0.f, 5, 6, 7), PT_C(1.364f, PT_R, PT_PI_F, 6, 7, 8),
             PT_C(1.409f, PT_R, 0.f, 6, 7, 8), PT_C(1.4f, PT_R, 0.f, 7, 8, 10)},
};
