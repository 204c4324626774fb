// CPU-only check of the host-side arithmetic of the C++ mirror (include/zkhip.hpp): the 4 x 64 Montgomery code for both BN254 fields,
// Fq2 square roots and the G2 point compression of `SerdeFormat::Processed`, and the small-degree interpolation of the SHPLONK prover.
// Built under ASan + UBSan by tests/test_host_arith.py, which feeds it points and expectations computed with Python integers.
//   usage: mirror_host_check <in.bin> <out.bin>
//   in : u32 m, m x G2Affine (128 B, Montgomery limbs); u32 d, d x (Fr point, Fr eval)   (Montgomery limbs)
//   out: per layout (0, 1): m x 64 compressed bytes; then per layout m x 128 bytes g2_decompress(g2_compress(P)); then d Fr coefficients
//        of the interpolation polynomial; then u64 flags (bit 0: Fq sqrt(x^2) squares back for 32 values, bit 1: Fq2 likewise,
//        bit 2: a non-square is refused, bit 3: Fr add / sub / neg / invert identities)
#include <cstdio>
#include <cstring>
#include <vector>
#include "zkhip.hpp"

using namespace zkhip::halo2;
namespace D = zkhip::halo2::detail;

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* in = fopen(argv[1], "rb");
  if (!in) return 2;
  uint32_t m = 0, d = 0;
  if (fread(&m, 4, 1, in) != 1) return 2;
  std::vector<std::array<uint64_t, 16>> pts(m);
  for (auto& p : pts) if (fread(p.data(), 8, 16, in) != 16) return 2;
  if (fread(&d, 4, 1, in) != 1) return 2;
  std::vector<Fr> xs(d), ys(d);
  for (uint32_t i = 0; i < d; i++) if (fread(&xs[i], 32, 1, in) != 1 || fread(&ys[i], 32, 1, in) != 1) return 2;
  fclose(in);
  FILE* out = fopen(argv[2], "wb");
  if (!out) return 2;
  for (int layout = 0; layout < 2; layout++)
    for (const auto& p : pts) { const auto c = D::g2_compress(p, layout); fwrite(c.data(), 1, 64, out); }
  for (int layout = 0; layout < 2; layout++)
    for (const auto& p : pts) {
      const auto c = D::g2_compress(p, layout);
      const auto back = D::g2_decompress(c.data(), layout);
      fwrite(back.data(), 8, 16, out);
    }
  const std::vector<Fr> coeffs = D::lagrange_interpolate(xs, ys);
  fwrite(coeffs.data(), 32, coeffs.size(), out);
  uint64_t flags = 0;
  {
    bool ok = true, ok2 = true;
    D::Fq x = D::fq_from_u64(0x1234567);
    for (int i = 0; i < 32; i++) {
      x = D::fq_add(D::fq_mul(x, x), D::fq_from_u64(7 + i));
      const D::Fq sq = D::fq_mul(x, x);
      D::Fq r;
      ok = ok && D::fq_sqrt(sq, &r) && D::fq_mul(r, r) == sq;
      const D::Fq2 a{x, D::fq_add(x, D::fq_from_u64(i))}, a2 = D::f2_mul(a, a);
      D::Fq2 r2;
      ok2 = ok2 && D::f2_sqrt(a2, &r2) && D::f2_mul(r2, r2) == a2;
    }
    if (ok) flags |= 1;
    if (ok2) flags |= 2;
    D::Fq r;
    if (!D::fq_sqrt(D::fq_from_u64(3), &r)) flags |= 4;          // 3 is not a square mod q (x = 0 is on no curve point)
    const Fr a = D::from_u64(123456789), b = D::from_u64(987654321);
    const bool fr_ok = D::sub_fr(D::add_fr(a, b), b) == a && D::add_fr(a, D::neg_fr(a)) == Fr{} && D::mul(a, D::invert(a)) == D::one() &&
                       D::add_fr(D::neg_fr(D::one()), D::one()) == Fr{};
    if (fr_ok) flags |= 8;
    // g2_is_valid (the RawBytes reader's check of g2 / s_g2): every input point passes, a point with one coordinate bumped does not,
    // a non-canonical limb pattern does not
    bool valid_ok = true;
    for (const auto& p : pts) {
      valid_ok = valid_ok && D::g2_is_valid(p);
      bool zero = true;
      for (uint64_t w : p) zero = zero && w == 0;
      if (zero) continue;
      auto bad = p;
      bad[8] ^= 1;                                               // y.c0 off by one limb bit: not on the twist
      valid_ok = valid_ok && !D::g2_is_valid(bad);
      auto big = p;
      big[3] = ~(uint64_t)0;                                     // x.c0 >= q: not a canonical residue
      valid_ok = valid_ok && !D::g2_is_valid(big);
    }
    if (valid_ok) flags |= 16;
  }
  fwrite(&flags, 8, 1, out);
  fclose(out);
  printf("mirror host check done\n");
  return 0;
}
