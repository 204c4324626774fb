// Drives the C++ mirror of keygen / key files / G2 / sharded commits (include/zkhip.hpp) the way a compiled host of the reference would;
// run by tests/test_gpu_prover_flow.py::test_cpp_keygen_mirror_matches_python, which builds the same circuit through the Python mirror and
// compares the two proving-key files byte for byte.
//   usage: keygen_driver <in.bin> <pk_out.bin> <report.bin> [pk_to_read.bin [pk_processed_out.bin params_raw_in.bin params_processed_out.bin]]
//   in : u32 k, u32 F (fixed columns), u32 P (permutation columns), u64 trapdoor, F x 2^k Fr, u32 copies, copies x (u32 lc, lr, rc, rr),
//        u32 m (G2 points), m x G2Affine, m x Fr
//   report: u64 flags (bit 0: ProvingKey::read(write(pk)) == pk, bit 1: the key file given as argv[4] reads back equal, bit 2: a truncated file
//           is refused, bit 3: commits agree for 1 and 5 MSM shards, bit 4: the key written with SerdeFormat::Processed reads back equal,
//           bit 5: the RawBytes parameter file given as argv[6], rewritten as Processed, reads back with the same points), then the G2 MSM
//           result (G2, 192 bytes)
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <vector>
#include "zkhip.hpp"

using namespace zkhip::halo2;

static bool same_key(const ProvingKey& a, const ProvingKey& b) {
  auto eqv = [](const std::vector<Fr>& x, const std::vector<Fr>& y) { return x.size() == y.size() && (x.empty() || !std::memcmp(x.data(), y.data(), x.size() * sizeof(Fr))); };
  auto eqs = [&](const std::vector<std::vector<Fr>>& x, const std::vector<std::vector<Fr>>& y) {
    if (x.size() != y.size()) return false;
    for (size_t i = 0; i < x.size(); i++) if (!eqv(x[i], y[i])) return false;
    return true;
  };
  auto eqp = [](const std::vector<G1Affine>& x, const std::vector<G1Affine>& y) { return x.size() == y.size() && (x.empty() || !std::memcmp(x.data(), y.data(), x.size() * sizeof(G1Affine))); };
  return a.vk.k == b.vk.k && eqp(a.vk.fixed_commitments, b.vk.fixed_commitments) && eqp(a.vk.permutation_commitments, b.vk.permutation_commitments) &&
         eqv(a.l0, b.l0) && eqv(a.l_last, b.l_last) && eqv(a.l_active_row, b.l_active_row) && eqs(a.fixed_values, b.fixed_values) &&
         eqs(a.fixed_polys, b.fixed_polys) && eqs(a.fixed_cosets, b.fixed_cosets) && eqs(a.permutations, b.permutations) &&
         eqs(a.permutation_polys, b.permutation_polys) && eqs(a.permutation_cosets, b.permutation_cosets);
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  FILE* in = fopen(argv[1], "rb");
  if (!in) return 2;
  uint32_t hdr[3];
  uint64_t trapdoor;
  if (fread(hdr, 4, 3, in) != 3 || fread(&trapdoor, 8, 1, in) != 1) return 2;
  const uint32_t k = hdr[0], F = hdr[1], P = hdr[2];
  const size_t n = (size_t)1 << k;
  std::vector<std::vector<Fr>> fixed(F, std::vector<Fr>(n));
  for (auto& col : fixed) if (fread(col.data(), sizeof(Fr), n, in) != n) return 2;
  uint32_t ncopies;
  if (fread(&ncopies, 4, 1, in) != 1) return 2;
  std::vector<uint32_t> copies(4 * (size_t)ncopies);
  if (ncopies && fread(copies.data(), 4, copies.size(), in) != copies.size()) return 2;
  uint32_t m;
  if (fread(&m, 4, 1, in) != 1) return 2;
  std::vector<G2Affine> g2pts(m);
  std::vector<Fr> g2sc(m);
  if (m && (fread(g2pts.data(), sizeof(G2Affine), m, in) != m || fread(g2sc.data(), sizeof(Fr), m, in) != m)) return 2;
  fclose(in);
  try {
    init({0});
    uint64_t flags = 0;
    CircuitShape cs;
    cs.num_fixed = F; cs.num_permutation_columns = P; cs.degree = 4; cs.blinding_factors = 5;
    Assembly assembly(n, P);
    for (uint32_t i = 0; i < ncopies; i++) assembly.copy(copies[4 * i], copies[4 * i + 1], copies[4 * i + 2], copies[4 * i + 3]);
    ParamsKZG params = ParamsKZG::setup(k, detail::from_u64(trapdoor));
    const VerifyingKey vk = keygen_vk(params, cs, fixed, assembly);
    const ProvingKey pk = keygen_pk(params, vk, cs, fixed, assembly);
    {
      std::ofstream f(argv[2], std::ios::binary);
      pk.write(f, SerdeFormat::RawBytesUnchecked);
    }
    {
      std::ifstream f(argv[2], std::ios::binary);
      if (same_key(ProvingKey::read(f, SerdeFormat::RawBytes, cs), pk)) flags |= 1;      // the checked reader: canonical scalars, points on the curve
    }
    if (argc > 4) {
      std::ifstream f(argv[4], std::ios::binary);
      if (same_key(ProvingKey::read(f, SerdeFormat::RawBytesUnchecked, cs), pk)) flags |= 2;
    }
    {
      std::stringstream whole;
      pk.write(whole);
      const std::string bytes = whole.str();
      std::stringstream cut(bytes.substr(0, bytes.size() - 3));
      try { ProvingKey::read(cut, SerdeFormat::RawBytesUnchecked, cs); } catch (const std::runtime_error&) { flags |= 4; }
    }
    {
      // the same commit with the SRS cut into 1 and into 5 point ranges (the multi-GPU path of the C ABI on one card)
      const std::vector<G1Affine> gl = params.get_g_lagrange();
      set_msm_shards(1);
      G1Affine one, five;
      {
        ParamsKZG p1(k, params.get_g(), gl);
        const G1 c = p1.commit_lagrange(fixed[0]);
        zkhip::halo2::check(zkhip_g1_batch_normalize(c.x, 1, one.x), "normalize");
      }
      set_msm_shards(5);
      {
        ParamsKZG p5(k, params.get_g(), gl);
        const G1 c = p5.commit_lagrange(fixed[0]);
        zkhip::halo2::check(zkhip_g1_batch_normalize(c.x, 1, five.x), "normalize");
      }
      set_msm_shards(0);
      if (!std::memcmp(&one, &five, sizeof(G1Affine)) && !std::memcmp(&one, &vk.fixed_commitments[0], sizeof(G1Affine))) flags |= 8;
    }
    if (argc > 7) {
      {
        std::ofstream f(argv[5], std::ios::binary);
        pk.write(f, SerdeFormat::Processed);
      }
      {
        std::ifstream f(argv[5], std::ios::binary);
        if (same_key(ProvingKey::read(f, SerdeFormat::Processed, cs), pk)) flags |= 16;
      }
      std::ifstream pf(argv[6], std::ios::binary);
      ParamsKZG raw = ParamsKZG::read(pf);
      {
        std::ofstream f(argv[7], std::ios::binary);
        raw.write_custom(f, SerdeFormat::Processed);
      }
      std::ifstream f(argv[7], std::ios::binary);
      ParamsKZG back = ParamsKZG::read_custom(f, SerdeFormat::Processed);
      if (back.k() == raw.k() && !std::memcmp(back.get_g().data(), raw.get_g().data(), raw.n() * sizeof(G1Affine)) &&
          !std::memcmp(back.get_g_lagrange().data(), raw.get_g_lagrange().data(), raw.n() * sizeof(G1Affine)) && back.g2() == raw.g2() && back.s_g2() == raw.s_g2())
        flags |= 32;
    }
    const G2 g2 = best_multiexp(g2sc, g2pts);
    FILE* rep = fopen(argv[3], "wb");
    fwrite(&flags, 8, 1, rep);
    fwrite(&g2, sizeof(G2), 1, rep);
    fclose(rep);
  } catch (const std::exception& e) {
    fprintf(stderr, "keygen_driver: %s\n", e.what());
    return 1;
  }
  return 0;
}
