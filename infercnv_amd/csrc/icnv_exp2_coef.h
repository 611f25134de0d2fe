// 2^f on [-0.5, 0.5] as a degree-11 polynomial in f: Chebyshev-node interpolation of 2^f computed in 60-digit
// arithmetic (coefficient c0 comes out as 1 - 3e-18 and is set to exactly 1); relative error of the polynomial
// itself 2.0e-17, of its Horner evaluation in fp64 fma arithmetic <= 1 ulp (tests/test_oracle.py::test_exp2_lean).
#pragma once
#define ICNV_EXP2_C1 0x1.62e42fefa39efp-1
#define ICNV_EXP2_C2 0x1.ebfbdff82c5aep-3
#define ICNV_EXP2_C3 0x1.c6b08d704a0c6p-5
#define ICNV_EXP2_C4 0x1.3b2ab6fb9f1a5p-7
#define ICNV_EXP2_C5 0x1.5d87fe78a3f9cp-10
#define ICNV_EXP2_C6 0x1.430913112c61bp-13
#define ICNV_EXP2_C7 0x1.ffcbfc6da6ed1p-17
#define ICNV_EXP2_C8 0x1.62bfc2c86d700p-20
#define ICNV_EXP2_C9 0x1.b524ebd13a55fp-24
#define ICNV_EXP2_C10 0x1.e6228acd1c6e5p-28
#define ICNV_EXP2_C11 0x1.e9ec1fcb69a7fp-32
