// C entry points of the KZG multi-open provers for a host that keeps its polynomials as device addresses of its own (the Rust shim's
// `DevCols`, rust-shim/prover_patch.rs): `ProverGWC::create_proof` (the reference's gen_snark path,
// /root/reference/aggregator/src/wrapper.rs:59-60, 127-137) and `ProverSHPLONK::create_proof` (the benches' gen_proof path,
// /root/reference/aggregator/benches/wrapper_circuit.rs:140) [DEP halo2-axiom poly/kzg/multiopen/{gwc, shplonk}/prover.rs].
//
// Host logic only: the provers are the C++ mirror of include/zkhip.hpp (rotation sets, interpolation of a handful of points, challenge
// powers on the host; linear combinations as fused row programs, divisions by (X - z) and the commitments on the device through the C ABI
// of this same library).  This file wraps them behind `extern "C"`: borrowed device addresses instead of DeviceVec objects, commitments
// against a base array registered with zkhip_register_bases (what `ParamsKZG::g` is under the shim), exceptions turned into status codes.
// SHPLONK comes in two calls because the transcript sits between them: H is written, then u is squeezed.
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <vector>
#include "../../include/zkhip.hpp"
#include "zkhip_internal.hpp"

using namespace zkhip::halo2;

namespace {

// commitments of 2^k device-resident coefficients against the registered array `bases` (a 96-byte device slot kept for the results)
struct registered_commit {
  const uint64_t* bases;
  size_t n;
  void* d_out = nullptr;
  registered_commit(const uint64_t* b, size_t n_) : bases(b), n(n_) { check(zkhip_alloc(sizeof(G1), &d_out), "multiopen: result slot"); }
  ~registered_commit() { (void)zkhip_free(d_out); }
  registered_commit(const registered_commit&) = delete;
  registered_commit& operator=(const registered_commit&) = delete;
  G1 operator()(const void* d_coeffs) const {
    G1 out;
    check(zkhip_msm_g1_registered_device(bases, d_coeffs, n, d_out, nullptr), "multiopen: commit");
    check(zkhip_download(&out, d_out, sizeof(G1)), "multiopen: commit download");
    return out;
  }
};

// the caller's queries as the mirror's: one borrowed DeviceVec per distinct device address (the provers tell polynomials apart by identity)
struct query_set {
  std::map<const void*, std::unique_ptr<DeviceVec>> views;
  std::vector<ProverQuery> queries;
  query_set(const zkhip_prover_query* q, size_t count, size_t n) {
    queries.reserve(count);
    for (size_t i = 0; i < count; i++) {
      auto it = views.find(q[i].d_poly);
      if (it == views.end()) it = views.emplace(q[i].d_poly, std::unique_ptr<DeviceVec>(new DeviceVec(DeviceVec::borrow(q[i].d_poly, n)))).first;
      ProverQuery pq;
      std::memcpy(pq.point.l, q[i].point, 32);
      pq.poly = it->second.get();
      pq.has_eval = q[i].has_eval != 0;
      if (pq.has_eval) std::memcpy(pq.eval.l, q[i].eval, 32);
      queries.push_back(pq);
    }
  }
};

int args_ok(const uint64_t* bases, uint32_t k, const zkhip_prover_query* queries, size_t n_queries) {
  if (!bases || !queries || n_queries == 0 || k > 28) { zkhip::set_error("multiopen: bad argument (bases / queries null, no query, or k > 28)"); return ZKHIP_EINVAL; }
  for (size_t i = 0; i < n_queries; i++)
    if (!queries[i].d_poly) { zkhip::set_error("multiopen: query %zu has no polynomial", i); return ZKHIP_EINVAL; }
  return ZKHIP_OK;
}

// exceptions of the mirror -> status: check() failures carry the failing call's own message (zkhip_last_error) in their text
template <class Body>
int guarded(const char* what, Body&& body) {
  try {
    body();
    return ZKHIP_OK;
  } catch (const std::invalid_argument& e) {
    zkhip::set_error("%s: %s", what, e.what());
    return ZKHIP_EINVAL;
  } catch (const std::logic_error& e) {
    zkhip::set_error("%s: %s", what, e.what());
    return ZKHIP_EINVAL;
  } catch (const std::bad_alloc&) {
    zkhip::set_error("%s: out of host memory", what);
    return ZKHIP_ENOMEM;
  } catch (const std::exception& e) {
    const bool inconsistent = std::strstr(e.what(), "L(u) != 0") != nullptr;
    const std::string msg = e.what();      // (set_error overwrites the buffer e.what() may quote)
    zkhip::set_error("%s: %s", what, msg.c_str());
    return inconsistent ? ZKHIP_EINVAL : ZKHIP_EHIP;
  }
}

}  // namespace

struct zkhip_shplonk {
  std::unique_ptr<registered_commit> commit;
  std::unique_ptr<query_set> qs;
  std::unique_ptr<ShplonkProver> prover;
};

extern "C" {

int zkhip_multiopen_gwc_device(const uint64_t* bases, uint32_t k, const zkhip_prover_query* queries, size_t n_queries, const uint64_t v[4], uint64_t* out_points,
                               size_t capacity, size_t* n_out) {
  int rc = args_ok(bases, k, queries, n_queries);
  if (rc != ZKHIP_OK) return rc;
  if (!v || !out_points || !n_out) { zkhip::set_error("multiopen_gwc: null pointer"); return ZKHIP_EINVAL; }
  return guarded("multiopen_gwc", [&] {
    const size_t n = (size_t)1 << k;
    registered_commit commit(bases, n);
    query_set qs(queries, n_queries, n);
    Fr vv;
    std::memcpy(vv.l, v, 32);
    const std::vector<G1> w = gwc_create_proof([&commit](const void* d) { return commit(d); }, k, qs.queries, vv);
    *n_out = w.size();
    if (w.size() > capacity) throw std::invalid_argument("room for fewer commitments than there are distinct points");
    std::memcpy(out_points, w.data(), w.size() * sizeof(G1));
  });
}

int zkhip_multiopen_shplonk_begin_device(const uint64_t* bases, uint32_t k, const zkhip_prover_query* queries, size_t n_queries, const uint64_t y[4],
                                         const uint64_t v[4], uint64_t out_h[12], zkhip_shplonk** state) {
  int rc = args_ok(bases, k, queries, n_queries);
  if (rc != ZKHIP_OK) return rc;
  if (!y || !v || !out_h || !state) { zkhip::set_error("multiopen_shplonk_begin: null pointer"); return ZKHIP_EINVAL; }
  *state = nullptr;
  std::unique_ptr<zkhip_shplonk> st;
  rc = guarded("multiopen_shplonk_begin", [&] {
    const size_t n = (size_t)1 << k;
    st.reset(new zkhip_shplonk);
    st->commit.reset(new registered_commit(bases, n));
    st->qs.reset(new query_set(queries, n_queries, n));
    registered_commit* c = st->commit.get();
    st->prover.reset(new ShplonkProver([c](const void* d) { return (*c)(d); }, k));
    Fr yy, vv;
    std::memcpy(yy.l, y, 32);
    std::memcpy(vv.l, v, 32);
    const G1 h = st->prover->begin(st->qs->queries, yy, vv);
    std::memcpy(out_h, &h, sizeof(G1));
  });
  if (rc == ZKHIP_OK) *state = st.release();
  return rc;
}

int zkhip_multiopen_shplonk_finish_device(zkhip_shplonk* state, const uint64_t u[4], uint64_t out_hp[12]) {
  if (!state) { zkhip::set_error("multiopen_shplonk_finish: null state"); return ZKHIP_EINVAL; }
  std::unique_ptr<zkhip_shplonk> st(state);                      // released whatever happens below
  if (!u || !out_hp) { zkhip::set_error("multiopen_shplonk_finish: null pointer"); return ZKHIP_EINVAL; }
  return guarded("multiopen_shplonk_finish", [&] {
    Fr uu;
    std::memcpy(uu.l, u, 32);
    const G1 hp = st->prover->finish(uu);
    std::memcpy(out_hp, &hp, sizeof(G1));
  });
}

int zkhip_multiopen_shplonk_abort(zkhip_shplonk* state) {
  delete state;
  return ZKHIP_OK;
}

}  // extern "C"
