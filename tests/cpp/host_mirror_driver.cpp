// Drives the C++ host mirror (include/zkhip.hpp) the way a compiled host of the reference would use it; run by
// tests/test_gpu_parity.py::test_cpp_host_mirror, which supplies the inputs and checks every output against the oracle.
//   usage: host_mirror_driver <in.bin> <out.bin>
//   in : u32 j, u32 k, then 2^k Fr (polynomial), 2^k G1Affine (SRS g)
//   out: omega, extended_omega (Fr each), commit (G1), best_fft(poly, omega), lagrange_to_coeff(poly), coeff_to_extended(poly),
//        divide_by_vanishing_poly(ext), extended_to_coeff(ext), eval_polynomial(poly, omega), kate_division(poly, omega),
//        grand_product, permute_expression_pair (2 vectors of 2^k - 6), a row program's output column,
//        setup(k, s): g[0..4), commit(poly), commit_lagrange(poly), ...; permutation_products (2 sets of 2^k)
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <sstream>
#include <vector>
#include "zkhip.hpp"

using namespace zkhip::halo2;

template <class T>
static void put(FILE* f, const std::vector<T>& v) { fwrite(v.data(), sizeof(T), v.size(), f); }

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  FILE* in = fopen(argv[1], "rb");
  if (!in) return 2;
  uint32_t jk[2];
  if (fread(jk, 4, 2, in) != 2) return 2;
  const uint32_t j = jk[0], k = jk[1];
  const size_t n = (size_t)1 << k;
  std::vector<Fr> poly(n);
  std::vector<G1Affine> g(n);
  if (fread(poly.data(), sizeof(Fr), n, in) != n || fread(g.data(), sizeof(G1Affine), n, in) != n) return 2;
  fclose(in);
  try {
    EvaluationDomain domain(j, k);
    ParamsKZG params(k, g);
    FILE* out = fopen(argv[2], "wb");
    fwrite(&domain.get_omega(), sizeof(Fr), 1, out);
    fwrite(&domain.get_extended_omega(), sizeof(Fr), 1, out);
    G1 c = params.commit(poly);
    fwrite(&c, sizeof(G1), 1, out);
    std::vector<Fr> f = poly;
    best_fft(f, domain.get_omega(), k);
    put(out, f);
    put(out, domain.lagrange_to_coeff(poly));
    std::vector<Fr> ext = domain.coeff_to_extended(poly);
    put(out, ext);
    put(out, domain.divide_by_vanishing_poly(ext));
    put(out, domain.extended_to_coeff(ext));
    Fr e = eval_polynomial(poly, domain.get_omega());
    fwrite(&e, sizeof(Fr), 1, out);
    put(out, kate_division(poly, domain.get_omega()));
    // 8(f): grand product of (poly / shifted poly), lookup permutation of a range-shaped pair, and a row program on device columns:
    // out[row] = poly[row] * poly[row + 1] + 7 * poly[row]
    std::vector<Fr> den(poly.begin() + 1, poly.end());
    den.push_back(poly[0]);
    put(out, grand_product(poly, den));
    std::vector<Fr> lin(n), ltab(n);
    for (size_t i = 0; i < n; i++) {
      lin[i] = detail::from_u64((poly[i].l[0] >> 7) % 37);
      ltab[i] = detail::from_u64(i % 37);
    }
    auto perm = permute_expression_pair(lin, ltab, n - 6);
    put(out, perm.first);
    put(out, perm.second);
    {
      DeviceVec col(poly), res(n);
      RowProgram rp;
      rp.constants.push_back(detail::from_u64(7));
      rp.rotations = {0, 1};
      rp.emit(ZKHIP_OP_MUL, 0, RowProgram::column(0, 0), RowProgram::constant(0));
      rp.emit(ZKHIP_OP_MAD, 1, RowProgram::column(0, 0), RowProgram::column(0, 1), RowProgram::reg(0));
      rp.result_reg = 1;
      rp.run({&col}, k, res);
      put(out, res.to_host());
    }
    // ParamsKZG::setup with a known trapdoor: first SRS points, and the same polynomial committed in both bases
    {
      ParamsKZG ps = ParamsKZG::setup(k, detail::from_u64(0x1234567ULL));
      fwrite(ps.get_g().data(), sizeof(G1Affine), 4, out);
      G1 c1 = ps.commit(poly), c2 = ps.commit_lagrange(poly);
      fwrite(&c1, sizeof(G1), 1, out);
      fwrite(&c2, sizeof(G1), 1, out);
      // ParamsKZG::write -> read: the same bytes come back, and the re-read parameters commit to the same point
      // real G2 points (the EIP-197 generator and its double, Montgomery limbs): the RawBytes reader checks g2 / s_g2 against the twist
      const ParamsKZG::G2Bytes g2 = {0x8e83b5d102bc2026ULL, 0xdceb1935497b0172ULL, 0xfbb8264797811adfULL, 0x19573841af96503bULL, 0xafb4737da84c6140ULL, 0x6043dd5a5802d8c4ULL,
                                     0x09e950fc52a02f86ULL, 0x14fef0833aea7b6bULL, 0x619dfa9d886be9f6ULL, 0xfe7fd297f59e9b78ULL, 0xff9e1a62231b7dfeULL, 0x28fd7eebae9e4206ULL,
                                     0x64095b56c71856eeULL, 0xdc57f922327d3cbbULL, 0x55f935be33351076ULL, 0x0da4a0e693fd6482ULL};
      const ParamsKZG::G2Bytes s_g2 = {0x42d3ae372af7a579ULL, 0xd2f609cbc0a293b5ULL, 0x57c1c76166f89cedULL, 0x2ee858391ef2cc42ULL, 0xa856e093920a72b8ULL, 0x63806a546d033f3eULL,
                                       0x1e7eeefa55543decULL, 0x056ca0f5743c0a90ULL, 0xbcd0a4b05c092587ULL, 0x5977f684ad4ff6baULL, 0x4fde1ced78e96a59ULL, 0x02750f99cf18d82eULL,
                                       0x332573d63fab5b02ULL, 0xa83429e5be54f062ULL, 0xaee376c3d20111f4ULL, 0x0e103b8b9272ef7dULL};
      ps.set_g2(g2, s_g2);
      std::stringstream file;
      ps.write(file);
      const std::string bytes = file.str();
      ParamsKZG again = ParamsKZG::read(file);
      std::stringstream file2;
      again.write(file2);
      const uint64_t same = (bytes == file2.str() && bytes.size() == 4 + 2 * n * sizeof(G1Affine) + 256 && again.s_g2() == s_g2) ? 1 : 0;
      fwrite(&same, 8, 1, out);
      G1 c3 = again.commit_lagrange(poly);
      fwrite(&c3, sizeof(G1), 1, out);
      bool threw_trunc = false;
      std::stringstream cut(bytes.substr(0, bytes.size() - 1));
      try { ParamsKZG::read(cut); } catch (const std::runtime_error&) { threw_trunc = true; }
      std::string forged = bytes;
      forged[bytes.size() - 256 + 64] ^= 1;                       // one bit of g2's y.c0: no longer on the twist
      bool threw_g2 = false, unchecked_ok = false;
      { std::stringstream f(forged); try { ParamsKZG::read(f); } catch (const std::runtime_error&) { threw_g2 = true; } }
      { std::stringstream f(forged); try { ParamsKZG::read(f, false); unchecked_ok = true; } catch (const std::runtime_error&) {} }
      const uint64_t trunc = (threw_trunc && threw_g2 && unchecked_ok) ? 1 : 0;
      fwrite(&trunc, 8, 1, out);
      // g_to_lagrange(g) reproduces the Lagrange basis that setup built from the trapdoor
      const std::vector<G1Affine> gl = g_to_lagrange(ps.get_g(), k);
      const uint64_t same_gl = std::memcmp(gl.data(), ps.get_g_lagrange().data(), n * sizeof(G1Affine)) == 0 ? 1 : 0;
      fwrite(&same_gl, 8, 1, out);
    }
    // permutation argument: 3 columns in sets of 2 -- columns = rotations of the polynomial's values, sigmas = other rotations; beta / gamma = omega / omega^2
    {
      std::vector<std::vector<Fr>> vals(3, poly), sig(3, poly);
      for (size_t c = 0; c < 3; c++) {
        std::rotate(vals[c].begin(), vals[c].begin() + (c + 1), vals[c].end());
        std::rotate(sig[c].begin(), sig[c].begin() + 5 * (c + 1), sig[c].end());
      }
      const Fr beta = domain.get_omega(), gamma = detail::from_u64(12345);
      for (const auto& z : permutation_products(vals, sig, 2, k, n - 6, beta, gamma, fr_delta(), domain.get_omega())) put(out, z);
    }
    fclose(out);
    // error behaviour: the reference's assert_eq!(coeffs.len(), bases.len())
    bool threw = false;
    try { best_multiexp(poly.data(), n, g.data(), n - 1); } catch (const std::invalid_argument&) { threw = true; }
    if (!threw) return 3;
  } catch (const std::exception& e) {
    fprintf(stderr, "host_mirror_driver: %s\n", e.what());
    return 1;
  }
  return 0;
}
