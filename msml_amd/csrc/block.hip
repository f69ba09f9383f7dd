// Block-level entry points: every launch of one IBasicBlock forward (backbones/frb/iresnet.py:56-67 of the reference,
// also the OSB encoder's blocks, backbones/osb/unet.py:80-91) enqueued by ONE call across the ABI.  The arithmetic is
// the sequence msml_amd/blocks.py issues launch by launch in the bf16 training path with accumulator-mode statistics --
// same kernels, same order, bit-identical results; what is saved is the host's per-launch work (argument conversion,
// wrapper logic: ~15 us per launch from Python, ~1300 launches per step).
//
//   o1 = bn1(x)                         msml_bn_fin_act_fwd  (statistics of x: accumulator `xacc`)
//   c1 = conv1(o1)                      msml_conv2d_acc      (statistics of c1 -> acc1)
//   o2 = prelu(bn2(c1))                 msml_bn_fin_act_fwd
//   c2 = conv2(o2)   (stride s)         msml_conv2d_acc      (-> acc2)
//   [d = conv_ds(x); idn = bn_ds(d)]    msml_conv2d_acc, msml_bn_fin_act_fwd      (first block of a stage)
//   out = bn3(c2) + idn                 msml_bn_fin_act_fwd  (statistics of out -> acc_out when the next block wants them)
#include "common.h"

enum {           // pointer table
  IB_X, IB_XACC,
  IB_BN1_G, IB_BN1_B, IB_BN1_RM, IB_BN1_RV, IB_COEF1, IB_O1,
  IB_WP1, IB_C1, IB_ACC1,
  IB_BN2_G, IB_BN2_B, IB_BN2_RM, IB_BN2_RV, IB_COEF2, IB_ALPHA, IB_O2,
  IB_WP2, IB_C2, IB_ACC2,
  IB_WPD, IB_D, IB_ACCD, IB_BND_G, IB_BND_B, IB_BND_RM, IB_BND_RV, IB_COEFD, IB_IDN,
  IB_BN3_G, IB_BN3_B, IB_BN3_RM, IB_BN3_RV, IB_COEF3, IB_OUT, IB_ACC_OUT,
  IB_NPTR
};
enum {           // int table
  II_N, II_H, II_W, II_CINP, II_COUTP, II_KOP1, II_KOP2, II_KOPD, II_STRIDE, II_P, II_Q, II_HAS_DS, II_DS_STRIDE, II_NINT
};
enum {           // float table: momentum, eps of bn1, bn2, bn3, downsample bn
  IF_MOM1, IF_EPS1, IF_MOM2, IF_EPS2, IF_MOM3, IF_EPS3, IF_MOMD, IF_EPSD, IF_NFLT
};

extern "C" int msml_iblock_fwd_tables(int* nptr, int* nint, int* nflt) {
  if (nptr) *nptr = IB_NPTR;
  if (nint) *nint = II_NINT;
  if (nflt) *nflt = IF_NFLT;
  return MSML_OK;
}

extern "C" int msml_iblock_fwd(const void* const* p, const int* ii, const float* ff, void* stream) {
  MSML_CHECK(p && ii && ff, MSML_ERR_SHAPE, "iblock_fwd: null table");
  const int N = ii[II_N], H = ii[II_H], W = ii[II_W], cin = ii[II_CINP], cout = ii[II_COUTP];
  const int P = ii[II_P], Q = ii[II_Q], stride = ii[II_STRIDE];
  const long m_in = (long)N * H * W, m_out = (long)N * P * Q;
  auto coef = [](const void* base, int i, int c) { return (float*)base + (long)i * c; };
  int rc;
  // bn1 (no activation) on x
  float* k1 = (float*)p[IB_COEF1];
  rc = msml_bn_fin_act_fwd((const double*)p[IB_XACC], (double)m_in, (const float*)p[IB_BN1_G], (const float*)p[IB_BN1_B],
                           (float*)p[IB_BN1_RM], (float*)p[IB_BN1_RV], ff[IF_MOM1], ff[IF_EPS1], coef(k1, 0, cin),
                           coef(k1, 1, cin), coef(k1, 2, cin), coef(k1, 3, cin), p[IB_X], nullptr, nullptr, 0,
                           (void*)p[IB_O1], m_in, cin, nullptr, MSML_BF16, stream);
  if (rc != MSML_OK) return rc;
  // conv1: 3x3 / stride 1 / pad 1 at the input resolution
  rc = msml_conv2d_acc(p[IB_O1], cin, nullptr, 0, p[IB_WP1], ii[II_KOP1], nullptr, (void*)p[IB_C1], cout,
                       (double*)p[IB_ACC1], N, H, W, H, W, 3, 3, 1, 1, 1, 0, MSML_BF16, MSML_BF16, stream);
  if (rc != MSML_OK) return rc;
  // bn2 + PReLU
  float* k2 = (float*)p[IB_COEF2];
  rc = msml_bn_fin_act_fwd((const double*)p[IB_ACC1], (double)m_in, (const float*)p[IB_BN2_G], (const float*)p[IB_BN2_B],
                           (float*)p[IB_BN2_RM], (float*)p[IB_BN2_RV], ff[IF_MOM2], ff[IF_EPS2], coef(k2, 0, cout),
                           coef(k2, 1, cout), coef(k2, 2, cout), coef(k2, 3, cout), p[IB_C1], (const float*)p[IB_ALPHA],
                           nullptr, 0, (void*)p[IB_O2], m_in, cout, nullptr, MSML_BF16, stream);
  if (rc != MSML_OK) return rc;
  // conv2 carries the stride
  rc = msml_conv2d_acc(p[IB_O2], cout, nullptr, 0, p[IB_WP2], ii[II_KOP2], nullptr, (void*)p[IB_C2], cout,
                       (double*)p[IB_ACC2], N, H, W, P, Q, 3, 3, stride, 1, 1, 0, MSML_BF16, MSML_BF16, stream);
  if (rc != MSML_OK) return rc;
  const void* idn = p[IB_X];
  if (ii[II_HAS_DS]) {
    rc = msml_conv2d_acc(p[IB_X], cin, nullptr, 0, p[IB_WPD], ii[II_KOPD], nullptr, (void*)p[IB_D], cout,
                         (double*)p[IB_ACCD], N, H, W, P, Q, 1, 1, ii[II_DS_STRIDE], 0, 0, 0, MSML_BF16, MSML_BF16, stream);
    if (rc != MSML_OK) return rc;
    float* kd = (float*)p[IB_COEFD];
    rc = msml_bn_fin_act_fwd((const double*)p[IB_ACCD], (double)m_out, (const float*)p[IB_BND_G], (const float*)p[IB_BND_B],
                             (float*)p[IB_BND_RM], (float*)p[IB_BND_RV], ff[IF_MOMD], ff[IF_EPSD], coef(kd, 0, cout),
                             coef(kd, 1, cout), coef(kd, 2, cout), coef(kd, 3, cout), p[IB_D], nullptr, nullptr, 0,
                             (void*)p[IB_IDN], m_out, cout, nullptr, MSML_BF16, stream);
    if (rc != MSML_OK) return rc;
    idn = p[IB_IDN];
  }
  // bn3 + identity (statistics of the block output for the next block's bn1 when acc_out is given)
  float* k3 = (float*)p[IB_COEF3];
  return msml_bn_fin_act_fwd((const double*)p[IB_ACC2], (double)m_out, (const float*)p[IB_BN3_G], (const float*)p[IB_BN3_B],
                             (float*)p[IB_BN3_RM], (float*)p[IB_BN3_RV], ff[IF_MOM3], ff[IF_EPS3], coef(k3, 0, cout),
                             coef(k3, 1, cout), coef(k3, 2, cout), coef(k3, 3, cout), p[IB_C2], nullptr, idn, 0,
                             (void*)p[IB_OUT], m_out, cout, (double*)p[IB_ACC_OUT], MSML_BF16, stream);
}
