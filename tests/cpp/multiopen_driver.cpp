// Drives the C++ mirror of the multi-open provers (include/zkhip.hpp: gwc_create_proof, shplonk_create_proof) on the inputs of
// tests/test_gpu_multiopen.py::test_cpp_multiopen_mirror_matches_python, which runs the Python mirror on the same inputs and compares the
// commitments in affine form.
//   usage: multiopen_driver <in.bin> <report.bin>
//   in : u32 k, u32 P (polynomials), u32 Q (queries), u64 trapdoor, P x 2^k Fr, Q x (u32 polynomial, Fr point), Fr y, Fr v, Fr u
//   report: u32 count of GWC witnesses, count x G1Affine, then H and H' of SHPLONK (2 x G1Affine), then u64 flags
//           (bit 0: a wrong evaluation makes shplonk_create_proof throw)
#include <cstdio>
#include <cstring>
#include <vector>
#include "zkhip.hpp"

using namespace zkhip::halo2;

static G1Affine affine(const G1& p) {
  G1Affine a;
  check(zkhip_g1_batch_normalize(p.x, 1, a.x), "normalize");
  return a;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* in = fopen(argv[1], "rb");
  if (!in) return 2;
  uint32_t hdr[3];
  uint64_t trapdoor;
  if (fread(hdr, 4, 3, in) != 3 || fread(&trapdoor, 8, 1, in) != 1) return 2;
  const uint32_t k = hdr[0], P = hdr[1], Q = hdr[2];
  const size_t n = (size_t)1 << k;
  std::vector<std::vector<Fr>> polys(P, std::vector<Fr>(n));
  for (auto& p : polys) if (fread(p.data(), sizeof(Fr), n, in) != n) return 2;
  std::vector<uint32_t> which(Q);
  std::vector<Fr> points(Q);
  for (uint32_t i = 0; i < Q; i++) if (fread(&which[i], 4, 1, in) != 1 || fread(&points[i], sizeof(Fr), 1, in) != 1 || which[i] >= P) return 2;
  Fr yvu[3];
  if (fread(yvu, sizeof(Fr), 3, in) != 3) return 2;
  fclose(in);
  try {
    init({0});
    ParamsKZG params = ParamsKZG::setup(k, detail::from_u64(trapdoor));
    DeviceCommitter committer(params.get_g());
    std::vector<DeviceVec> d_polys;
    for (const auto& p : polys) d_polys.emplace_back(p);
    std::vector<ProverQuery> queries;
    for (uint32_t i = 0; i < Q; i++) queries.push_back(ProverQuery{points[i], &d_polys[which[i]]});
    const std::vector<G1> W = gwc_create_proof(committer, k, queries, yvu[1]);
    const std::pair<G1, G1> HH = shplonk_create_proof(committer, k, queries, yvu[0], yvu[1], yvu[2]);
    uint64_t flags = 0;
    {
      // one evaluation supplied by the caller and off by one: the prover's own L(u) = 0 assertion must fire
      std::vector<ProverQuery> bad = queries;
      DeviceVec d_e(1);
      check(zkhip_fr_eval_polynomial_device(bad[1].poly->data(), n, bad[1].point.l, d_e.data(), nullptr), "eval");
      bad[1].eval = detail::add_fr(d_e.to_host()[0], detail::one());
      bad[1].has_eval = true;
      try { shplonk_create_proof(committer, k, bad, yvu[0], yvu[1], yvu[2]); } catch (const std::runtime_error&) { flags |= 1; }
    }
    FILE* rep = fopen(argv[2], "wb");
    const uint32_t count = (uint32_t)W.size();
    fwrite(&count, 4, 1, rep);
    for (const G1& w : W) { const G1Affine a = affine(w); fwrite(&a, sizeof a, 1, rep); }
    const G1Affine h = affine(HH.first), hp = affine(HH.second);
    fwrite(&h, sizeof h, 1, rep);
    fwrite(&hp, sizeof hp, 1, rep);
    fwrite(&flags, 8, 1, rep);
    fclose(rep);
  } catch (const std::exception& e) {
    fprintf(stderr, "multiopen_driver: %s\n", e.what());
    return 1;
  }
  return 0;
}
