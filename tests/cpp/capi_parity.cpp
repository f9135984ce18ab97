// tests/cpp/capi_parity.cpp — the C ABI driven from compiled code (no Python, no torch): seeded random
// LSDB-shaped graphs through include/holo_spf_hip.hpp, compared bit for bit with the CPU oracle
// (oracle/liboracle_spf.so, loaded with dlopen: TEST INFRASTRUCTURE, see oracle/spf_oracle.cpp), plus the
// error-code contract and the device-resident path (hspf_run_device + hspf_routes_device on hipMalloc'ed
// buffers).  Exit codes: 0 all checks passed, 77 no HIP device (the CPU-only build container), 1 mismatch.
//
// Build (done by __graft_entry__.build()):
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude tests/cpp/capi_parity.cpp -Lholo_amd -lholo_spf_hip
//         -Wl,-rpath,$ORIGIN/../../holo_amd -ldl -o tests/cpp/capi_parity
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "holo_spf_hip.hpp"

using u32 = uint32_t;
typedef int (*oracle_run_t)(u32, u32, const u32 *, const u32 *, const u32 *, const uint8_t *, u32, const u32 *, u32, u32, int,
                            u32 *, uint16_t *, uint16_t *, u32 *, uint64_t *, u32, u32 *, u32 *, uint64_t *);

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static u32 rnd(u32 m) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (u32)((rng_state >> 11) % m); }

struct Lsdb { std::vector<u32> row_ptr, col, metric; std::vector<uint8_t> vflags; u32 n; };

// networks [0, n_net) then routers: router-router p2p links (some one-way, some parallel), LANs with
// zero-cost network->router links, overloaded and non-expandable vertices.
static Lsdb make(u32 n_rtr, u32 n_net, u32 seed) {
  rng_state = 0x9E3779B97F4A7C15ull ^ (seed * 0x100000001B3ull);
  const u32 n = n_rtr + n_net;
  std::vector<std::vector<std::pair<u32, u32>>> rows(n);
  for (u32 i = 0; i < n_rtr * 3 / 2; ++i) {
    const u32 u = n_net + rnd(n_rtr), v = n_net + rnd(n_rtr);
    if (u == v) continue;
    rows[u].push_back({v, 1 + rnd(6)});
    if (rnd(100) >= 3) rows[v].push_back({u, 1 + rnd(6)});
    if (rnd(100) < 5) { rows[u].push_back({v, 1 + rnd(6)}); rows[v].push_back({u, 1 + rnd(6)}); }
  }
  for (u32 net = 0; net < n_net; ++net)
    for (u32 k = 0; k < 4; ++k) {
      const u32 r = n_net + rnd(n_rtr);
      rows[r].push_back({net, 1 + rnd(6)});
      rows[net].push_back({r, 0});
    }
  Lsdb g; g.n = n; g.row_ptr.assign(n + 1, 0); g.vflags.assign(n, 0);
  for (u32 u = 0; u < n; ++u) {
    for (auto &e : rows[u]) { g.col.push_back(e.first); g.metric.push_back(e.second); }
    g.row_ptr[u + 1] = (u32)g.col.size();
    if (u < n_net) g.vflags[u] |= HSPF_VF_NETWORK;
    else if (rnd(100) < 3) g.vflags[u] |= HSPF_VF_NO_TRANSIT;
    if (rnd(100) < 2) g.vflags[u] |= HSPF_VF_NO_EXPAND;
  }
  return g;
}

#define CHECK(c, msg) do { if (!(c)) { std::fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, msg); return 1; } } while (0)

int main(int argc, char **argv) {
  if (hspf_device_count() <= 0) {
    hspf_ctx *c = nullptr;
    const int rc = hspf_init(0, &c);
    std::printf("no HIP device: hspf_init -> %d (%s)\n", rc, hspf_strerror(rc));
    return (rc == HSPF_E_NODEV && c == nullptr) ? 77 : 1;
  }
  const std::string here = argc > 1 ? argv[1] : ".";
  void *so = dlopen((here + "/oracle/liboracle_spf.so").c_str(), RTLD_NOW);
  CHECK(so, "dlopen oracle/liboracle_spf.so (run `make -C oracle`)");
  auto oracle = (oracle_run_t)dlsym(so, "oracle_spf_run");
  CHECK(oracle, "oracle_spf_run");

  hspf::Engine eng(0);
  int checked = 0, packed = 0;
  for (u32 seed = 0; seed < 6; ++seed)
    for (u32 flags : {0u, (u32)HSPF_RUN_NET_NEXTHOPS, (u32)HSPF_RUN_IGNORE_OVERLOAD}) {
      Lsdb g = make(120 + 37 * seed, 10 + seed, seed);
      hspf::Graph G = eng.upload(g.row_ptr, g.col, g.metric, g.vflags, 0xFE000000u);
      std::vector<u32> roots;
      for (u32 r = 10 + seed; r < 10 + seed + 90 && r < g.n; ++r) roots.push_back(r);
      roots[3] = HSPF_NO_ROOT;                                   // padding entry: empty SPT
      hspf::Tables t = eng.run(G, roots, flags);
      const size_t rn = (size_t)roots.size() * g.n;
      std::vector<u32> d(rn), pr(rn), nn(rn), np(rn); std::vector<uint16_t> h(rn), f(rn); std::vector<uint64_t> m(rn * t.mask_words), wk(roots.size());
      const int rc = oracle(g.n, (u32)g.col.size(), g.row_ptr.data(), g.col.data(), g.metric.data(), g.vflags.data(), 0xFE000000u,
                            roots.data(), (u32)roots.size(), flags, /*variant MAP*/ 1, d.data(), h.data(), f.data(), pr.data(), m.data(),
                            t.mask_words, nn.data(), np.data(), wk.data());
      CHECK(rc == 0, "oracle failed");
      CHECK(d == t.dist, "dist");
      CHECK(h == t.hops, "hops");
      for (size_t i = 0; i < rn; ++i) CHECK((t.flags[i] & 1) == f[i], "in-SPT flag");
      CHECK(m == t.mask, "first-hop mask");
      ++checked;
    }

  // error contract: codes, not crashes
  {
    Lsdb g = make(50, 4, 99);
    hspf::Graph G = eng.upload(g.row_ptr, g.col, g.metric, g.vflags, 0xFE000000u);
    u32 bad = g.n + 7, dist = 0;
    hspf_result out{&dist, nullptr, nullptr, nullptr, 1, nullptr};
    CHECK(hspf_run(eng.raw(), G.raw(), &bad, 1, 0, &out) == HSPF_E_INVAL, "root out of range must be HSPF_E_INVAL");
    CHECK(hspf_run(eng.raw(), G.raw(), nullptr, 1, 0, &out) == HSPF_E_INVAL, "NULL roots");
    std::vector<u32> col2 = g.col; col2[0] = g.n + 1;
    hspf_csr csr{g.n, (u32)col2.size(), g.row_ptr.data(), col2.data(), g.metric.data(), g.vflags.data(), 0xFE000000u};
    hspf_graph *gg = nullptr;
    CHECK(hspf_graph_upload(eng.raw(), &csr, &gg) == HSPF_E_INVAL && gg == nullptr, "col out of range");
  }

  // ABI 6: several runs in flight on the lanes of the one context (hspf_run_device_async / hspf_wait) against the synchronous
  // host-buffer results of the same root sets — from compiled code, plain hipMalloc buffers
  int async_checked = 0;
  {
    Lsdb g = make(260, 14, 21);
    hspf::Graph G = eng.upload(g.row_ptr, g.col, g.metric, g.vflags, 0xFE000000u);
    const u32 n = g.n, K = 5;
    std::vector<std::vector<u32>> sets(K);
    for (u32 k = 0; k < K; ++k) for (u32 r = 0; r < 70; ++r) sets[k].push_back((14 + 31 * k + r) % n);
    u32 W = 1;
    for (auto &rs : sets) W = std::max(W, eng.mask_words(G, rs));
    struct Dev { u32 *d; uint16_t *h, *f; uint64_t *m; };
    std::vector<Dev> bufs(K);
    std::vector<uint64_t> tickets;
    CHECK(eng.async_lanes() >= 1, "hspf_async_lanes");
    for (u32 k = 0; k < K; ++k) {
      const size_t rn = (size_t)sets[k].size() * n;
      hipMalloc(&bufs[k].d, rn * 4); hipMalloc(&bufs[k].h, rn * 2); hipMalloc(&bufs[k].f, rn * 2); hipMalloc(&bufs[k].m, rn * 8 * W);
      hspf_result od{bufs[k].d, bufs[k].h, bufs[k].f, bufs[k].m, W, nullptr};
      tickets.push_back(eng.run_device_async(G, sets[k], HSPF_RUN_NET_NEXTHOPS, od));
    }
    for (u32 k = 0; k < K; ++k) {
      const hspf_stats st = eng.wait(tickets[k]);
      CHECK(st.n_roots == sets[k].size(), "stats of the ticket");
      hspf::Tables t = eng.run(G, sets[k], HSPF_RUN_NET_NEXTHOPS);
      const size_t rn = (size_t)sets[k].size() * n;
      std::vector<u32> d(rn); std::vector<uint16_t> h(rn), f(rn); std::vector<uint64_t> m(rn * W);
      hipMemcpy(d.data(), bufs[k].d, rn * 4, hipMemcpyDeviceToHost); hipMemcpy(h.data(), bufs[k].h, rn * 2, hipMemcpyDeviceToHost);
      hipMemcpy(f.data(), bufs[k].f, rn * 2, hipMemcpyDeviceToHost); hipMemcpy(m.data(), bufs[k].m, rn * 8 * W, hipMemcpyDeviceToHost);
      CHECK(d == t.dist && h == t.hops, "async dist / hops");
      for (size_t i = 0; i < rn; ++i) CHECK((f[i] & 1) == (t.flags[i] & 1), "async in-SPT flag");
      for (size_t i = 0; i < rn; ++i) for (u32 w = 0; w < t.mask_words; ++w) CHECK(m[i * W + w] == t.mask[i * t.mask_words + w], "async first-hop mask");
      ++async_checked;
      hipFree(bufs[k].d); hipFree(bufs[k].h); hipFree(bufs[k].f); hipFree(bufs[k].m);
    }
    uint64_t bad_ticket = 1ull << 40;
    CHECK(hspf_wait(eng.raw(), bad_ticket, nullptr) == HSPF_E_INVAL, "a ticket never handed out is HSPF_E_INVAL");
    CHECK(hspf::Engine::recommend_cpu(25, 80, 1) && !hspf::Engine::recommend_cpu(100000, 1000000, 1), "hspf_recommend_cpu");
  }

  // device-resident path: hspf_run_device + hspf_routes_device on plain hipMalloc buffers
  {
    Lsdb g = make(200, 12, 5);
    hspf::Graph G = eng.upload(g.row_ptr, g.col, g.metric, g.vflags, 0xFE000000u);
    std::vector<u32> roots; for (u32 r = 12; r < 12 + 70; ++r) roots.push_back(r);
    const u32 W = eng.mask_words(G, roots), R = (u32)roots.size(), n = g.n;
    hspf::Tables t = eng.run(G, roots, 0);
    u32 *dd; uint16_t *dh, *df; uint64_t *dm;
    hipMalloc(&dd, (size_t)R * n * 4); hipMalloc(&dh, (size_t)R * n * 2); hipMalloc(&df, (size_t)R * n * 2); hipMalloc(&dm, (size_t)R * n * 8 * W);
    hspf_result od{dd, dh, df, dm, W, nullptr};
    CHECK(hspf_run_device(eng.raw(), G.raw(), roots.data(), R, 0, &od) == HSPF_OK, "hspf_run_device");
    // prefix table: prefix p advertised by vertices p and (7p+3) mod n with metrics 1 and 2
    const u32 P = n;
    std::vector<u32> pptr(P + 1), pv, pm;
    for (u32 p = 0; p < P; ++p) { u32 a = p, b = (7 * p + 3) % n; if (a > b) std::swap(a, b); pv.push_back(a); pm.push_back(a == p ? 1 : 2); if (b != a) { pv.push_back(b); pm.push_back(b == p ? 1 : 2); } pptr[p + 1] = (u32)pv.size(); }
    u32 *bm, *be; uint64_t *nm;
    hipMalloc(&bm, (size_t)R * P * 4); hipMalloc(&be, (size_t)R * P * 4); hipMalloc(&nm, (size_t)R * P * 8 * W);
    hspf_prefix_table tab{P, (u32)pv.size(), pptr.data(), pv.data(), pm.data(), 0};
    hspf_routes ro{bm, be, nm};
    CHECK(hspf_routes_device(eng.raw(), n, R, W, dd, df, dm, &tab, &ro) == HSPF_OK, "hspf_routes_device");
    std::vector<u32> hbm((size_t)R * P), hbe((size_t)R * P); std::vector<uint64_t> hnm((size_t)R * P * W);
    hipMemcpy(hbm.data(), bm, hbm.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hbe.data(), be, hbe.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hnm.data(), nm, hnm.size() * 8, hipMemcpyDeviceToHost);
    for (u32 r = 0; r < R; ++r)
      for (u32 p = 0; p < P; ++p) {
        u32 best = 0xFFFFFFFFu, ent = 0xFFFFFFFFu; std::vector<uint64_t> acc(W, 0);
        for (u32 e = pptr[p]; e < pptr[p + 1]; ++e) {
          const size_t i = (size_t)r * n + pv[e];
          if (!(t.flags[i] & 1)) continue;
          const u32 mm = t.dist[i] + pm[e];
          if (mm < best) { best = mm; ent = e; for (u32 w = 0; w < W; ++w) acc[w] = t.mask[i * W + w]; }
          else if (mm == best) for (u32 w = 0; w < W; ++w) acc[w] |= t.mask[i * W + w];
        }
        const size_t o = (size_t)r * P + p;
        CHECK(hbm[o] == best && hbe[o] == ent, "route metric / entry");
        for (u32 w = 0; w < W; ++w) CHECK(hnm[o * W + w] == acc[w], "route next-hop mask");
      }
    // HSPF_PFX_ORDERED (OSPFv3's order-dependent fold, holo-ospf/src/route.rs:343-448): the same entries in REVERSED
    // order inside each prefix, every second entry a network entry with an origin, an initial route for every third prefix
    {
      std::vector<u32> pv2(pv.size()), pm2(pm.size()), org(pv.size()), imet(P), iorg(P); std::vector<uint8_t> iex(P);
      for (u32 p = 0; p < P; ++p) {
        const u32 a = pptr[p], b = pptr[p + 1];
        for (u32 e = a; e < b; ++e) {
          const u32 src = a + (b - 1 - e);
          pv2[e] = pv[src] | ((e & 1u) ? HSPF_PFX_ENTRY_NETWORK : 0u); pm2[e] = pm[src]; org[e] = (7 * e + p) % 5;
        }
        iex[p] = (p % 3 == 0) ? 1 : 0; imet[p] = 3 + p % 9; iorg[p] = p % 5;
      }
      hspf_prefix_table tab2{P, (u32)pv2.size(), pptr.data(), pv2.data(), pm2.data(), HSPF_PFX_SATURATING | HSPF_PFX_ORDERED,
                             org.data(), iex.data(), imet.data(), iorg.data()};
      u32 *bm2, *be2; uint64_t *nm2;
      hipMalloc(&bm2, (size_t)R * P * 4); hipMalloc(&be2, (size_t)R * P * 4); hipMalloc(&nm2, (size_t)R * P * 8 * W);
      hspf_routes ro2{bm2, be2, nm2};
      CHECK(hspf_routes_device(eng.raw(), n, R, W, dd, df, dm, &tab2, &ro2) == HSPF_OK, "hspf_routes_device (ordered)");
      std::vector<u32> hb2((size_t)R * P), he2((size_t)R * P); std::vector<uint64_t> hn2((size_t)R * P * W);
      hipMemcpy(hb2.data(), bm2, hb2.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(he2.data(), be2, he2.size() * 4, hipMemcpyDeviceToHost);
      hipMemcpy(hn2.data(), nm2, hn2.size() * 8, hipMemcpyDeviceToHost);
      for (u32 r = 0; r < R; ++r)
        for (u32 p = 0; p < P; ++p) {
          bool ex = iex[p] != 0; u32 best = ex ? imet[p] : 0xFFFFFFFFu, bo = ex ? iorg[p] : 0, ent = ex ? HSPF_PFX_KEPT_INIT : 0xFFFFFFFFu;
          std::vector<uint64_t> acc(W, 0);
          for (u32 e = pptr[p]; e < pptr[p + 1]; ++e) {
            const u32 v = pv2[e] & 0x7FFFFFFFu; const size_t i = (size_t)r * n + v;
            if (!(t.flags[i] & 1)) continue;
            const uint64_t s64 = (uint64_t)t.dist[i] + pm2[e]; const u32 mm = s64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)s64;
            if (ex && mm > best) continue;
            if ((pv2[e] & HSPF_PFX_ENTRY_NETWORK) && ex) { if (mm < best || (mm == best && org[e] > bo)) ex = false; else continue; }
            if (!ex || mm < best) { ex = true; best = mm; bo = org[e]; ent = e; for (u32 w = 0; w < W; ++w) acc[w] = t.mask[i * W + w]; }
            else for (u32 w = 0; w < W; ++w) acc[w] |= t.mask[i * W + w];
          }
          const size_t o = (size_t)r * P + p;
          CHECK(hb2[o] == (ex ? best : 0xFFFFFFFFu) && he2[o] == ent, "ordered fold: metric / owner entry");
          for (u32 w = 0; w < W; ++w) CHECK(hn2[o * W + w] == acc[w], "ordered fold: next-hop mask");
        }
      // RIB diff (update_global_rib's comparison, holo-isis/src/route.rs:254-312): the IS-IS table against the ordered one
      uint8_t *act; u32 *chg, *cptr;
      hipMalloc(&act, (size_t)R * P); hipMalloc(&chg, (size_t)R * P * 4); hipMalloc(&cptr, ((size_t)R + 1) * 4);
      CHECK(hspf_routes_diff_device(eng.raw(), R, P, W, &ro, &ro2, act, chg, cptr) == HSPF_OK, "hspf_routes_diff_device");
      std::vector<uint8_t> ha((size_t)R * P); std::vector<u32> hc((size_t)R * P), hp(R + 1);
      hipMemcpy(ha.data(), act, ha.size(), hipMemcpyDeviceToHost); hipMemcpy(hc.data(), chg, hc.size() * 4, hipMemcpyDeviceToHost);
      hipMemcpy(hp.data(), cptr, hp.size() * 4, hipMemcpyDeviceToHost);
      u32 pos = 0;
      for (u32 r = 0; r < R; ++r) {
        CHECK(hp[r] == pos, "diff: changed_ptr");
        for (u32 p = 0; p < P; ++p) {
          const size_t o = (size_t)r * P + p;
          const bool had = hbe[o] != 0xFFFFFFFFu, has = he2[o] != 0xFFFFFFFFu;
          bool same = true, onh = false, nnh = false;
          for (u32 w = 0; w < W; ++w) { same = same && hnm[o * W + w] == hn2[o * W + w]; onh = onh || hnm[o * W + w]; nnh = nnh || hn2[o * W + w]; }
          u32 a = has ? ((had && hbm[o] == hb2[o] && same) ? HSPF_DIFF_SAME : (nnh ? HSPF_DIFF_INSTALL : HSPF_DIFF_SILENT))
                      : (had ? (onh ? HSPF_DIFF_WITHDRAW : HSPF_DIFF_SILENT) : HSPF_DIFF_SAME);
          CHECK(ha[o] == a, "diff: action");
          if (a == HSPF_DIFF_INSTALL || a == HSPF_DIFF_WITHDRAW) { CHECK(hc[pos] == p, "diff: compacted index"); ++pos; }
        }
      }
      CHECK(hp[R] == pos, "diff: total");
      // the hand-off: the changed routes as ONE record stream, one copy
      {
        const u32 nrec = hspf_routes_diff_count(eng.raw());
        CHECK(nrec == pos, "hspf_routes_diff_count");
        const u32 rw = HSPF_ROUTE_REC_WORDS + 2 * W;
        std::vector<u32> rec((size_t)nrec * rw);
        CHECK(hspf_routes_pack(eng.raw(), R, P, W, &ro2, act, chg, cptr, nrec, rec.data()) == HSPF_OK, "hspf_routes_pack");
        u32 k = 0;
        for (u32 r = 0; r < R; ++r)
          for (u32 q = hp[r]; q < hp[r + 1]; ++q, ++k) {
            const u32 p = hc[q]; const size_t o = (size_t)r * P + p; const u32 *x = &rec[(size_t)k * rw];
            CHECK(x[0] == r && x[1] == p && x[2] == ha[o] && x[3] == hb2[o] && x[4] == he2[o] && x[5] == 0, "packed record header");
            for (u32 w = 0; w < W; ++w)
              CHECK((((uint64_t)x[HSPF_ROUTE_REC_WORDS + 2 * w + 1] << 32) | x[HSPF_ROUTE_REC_WORDS + 2 * w]) == hn2[o * W + w], "packed record mask");
          }
        CHECK(k == nrec, "packed record count");
        ++packed;
      }
      // HSPF_PFX_RESIDENT: the caller's unchanged plain table again (uploaded afresh after the ordered one), then resident
      {
        u32 *bm3, *be3; uint64_t *nm3;
        hipMalloc(&bm3, (size_t)R * P * 4); hipMalloc(&be3, (size_t)R * P * 4); hipMalloc(&nm3, (size_t)R * P * 8 * W);
        hspf_routes ro3{bm3, be3, nm3};
        hspf_prefix_table tab3 = tab; tab3.flags |= HSPF_PFX_RESIDENT;
        for (int pass = 0; pass < 2; ++pass) {                     // pass 0: nothing recorded matches (the ordered table was last) -> upload; pass 1: resident
          hipMemset(bm3, 0xEE, (size_t)R * P * 4);
          hipDeviceSynchronize();                                  // (the engine's stream does not wait for the null stream)
          CHECK(hspf_routes_device(eng.raw(), n, R, W, dd, df, dm, &tab3, &ro3) == HSPF_OK, "hspf_routes_device (resident)");
          std::vector<u32> hb3((size_t)R * P), he3((size_t)R * P); std::vector<uint64_t> hn3((size_t)R * P * W);
          hipMemcpy(hb3.data(), bm3, hb3.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(he3.data(), be3, he3.size() * 4, hipMemcpyDeviceToHost);
          hipMemcpy(hn3.data(), nm3, hn3.size() * 8, hipMemcpyDeviceToHost);
          CHECK(hb3 == hbm && he3 == hbe && hn3 == hnm, "resident table: routes differ");
        }
        hipFree(bm3); hipFree(be3); hipFree(nm3);
      }
      hipFree(bm2); hipFree(be2); hipFree(nm2); hipFree(act); hipFree(chg); hipFree(cptr);
    }
    hipFree(dd); hipFree(dh); hipFree(df); hipFree(dm); hipFree(bm); hipFree(be); hipFree(nm);
  }
  // incremental update: rows replaced through hspf::Graph::patch, results against the oracle on the patched CSR
  int patched = 0, cost_patched = 0;
  {
    Lsdb g = make(150, 10, 21);
    hspf::Graph G = eng.upload(g.row_ptr, g.col, g.metric, g.vflags, 0xFE000000u);
    std::vector<u32> roots; for (u32 r = 10; r < 10 + 64; ++r) roots.push_back(r);
    for (u32 round = 0; round < 5; ++round) {
      // rewrite up to 1 + 3*round router rows: drop a third of the links, re-cost the rest, add one link
      std::vector<hspf::Graph::Row> rows;
      std::vector<std::vector<std::pair<u32, u32>>> all(g.n);
      for (u32 u = 0; u < g.n; ++u)
        for (u32 k = g.row_ptr[u]; k < g.row_ptr[u + 1]; ++k) all[u].push_back({g.col[k], g.metric[k]});
      for (u32 j = 0; j < 1 + 3 * round; ++j) {
        const u32 u = 10 + rnd(150);
        bool dup = false; for (auto &r : rows) dup |= r.vertex == u;
        if (dup) continue;
        std::vector<std::pair<u32, u32>> nr;
        for (auto &e : all[u]) if (rnd(3)) nr.push_back({e.first, e.first < 10 ? e.second : 1 + rnd(6)});
        nr.push_back({10 + rnd(150), 1 + rnd(6)});
        all[u] = nr;
        g.vflags[u] ^= (rnd(4) == 0) ? HSPF_VF_NO_TRANSIT : 0;
        hspf::Graph::Row row{u, {}, {}, g.vflags[u]};
        for (auto &e : nr) { row.col.push_back(e.first); row.metric.push_back(e.second); }
        rows.push_back(row);
      }
      G.patch(rows);
      g.col.clear(); g.metric.clear();
      for (u32 u = 0; u < g.n; ++u) {
        for (auto &e : all[u]) { g.col.push_back(e.first); g.metric.push_back(e.second); }
        g.row_ptr[u + 1] = (u32)g.col.size();
      }
      CHECK(G.n_links() == g.col.size(), "patched link count");
      hspf::Tables t = eng.run(G, roots, HSPF_RUN_NET_NEXTHOPS);
      const size_t rn = (size_t)roots.size() * g.n;
      std::vector<u32> d(rn), pr(rn), nn(rn), np(rn); std::vector<uint16_t> h(rn), f(rn); std::vector<uint64_t> m(rn * t.mask_words), wk(roots.size());
      const int rc = oracle(g.n, (u32)g.col.size(), g.row_ptr.data(), g.col.data(), g.metric.data(), g.vflags.data(), 0xFE000000u,
                            roots.data(), (u32)roots.size(), HSPF_RUN_NET_NEXTHOPS, 1, d.data(), h.data(), f.data(), pr.data(), m.data(),
                            t.mask_words, nn.data(), np.data(), wk.data());
      CHECK(rc == 0, "oracle failed");
      CHECK(d == t.dist && h == t.hops && m == t.mask, "patched graph differs from the oracle on the patched CSR");
      ++patched;
    }
    // metric-only refreshes: same targets, order and flags -> applied in place (build mode 2), results as the oracle's
    for (u32 round = 0; round < 4; ++round) {
      std::vector<hspf::Graph::Row> rows;
      for (u32 j = 0; j < 1 + 2 * round; ++j) {
        const u32 u = 10 + rnd(150);
        bool dup = false; for (auto &r : rows) dup |= r.vertex == u;
        if (dup) continue;
        hspf::Graph::Row row{u, {}, {}, g.vflags[u]};
        for (u32 k = g.row_ptr[u]; k < g.row_ptr[u + 1]; ++k) {
          if (g.col[k] >= 10) g.metric[k] = 1 + rnd(9);                          // links into a network keep their cost
          row.col.push_back(g.col[k]); row.metric.push_back(g.metric[k]);
        }
        rows.push_back(row);
      }
      G.patch(rows);
      u32 mode = 7; size_t got = 0;
      CHECK(hspf_graph_export(eng.raw(), G.raw(), HSPF_GX_BUILD_MODE, &mode, 4, &got) == HSPF_OK && mode == 2, "cost-only patch must be applied in place");
      hspf::Tables t = eng.run(G, roots, HSPF_RUN_NET_NEXTHOPS);
      const size_t rn = (size_t)roots.size() * g.n;
      std::vector<u32> d(rn), pr(rn), nn(rn), np(rn); std::vector<uint16_t> h(rn), f(rn); std::vector<uint64_t> m(rn * t.mask_words), wk(roots.size());
      CHECK(oracle(g.n, (u32)g.col.size(), g.row_ptr.data(), g.col.data(), g.metric.data(), g.vflags.data(), 0xFE000000u,
                   roots.data(), (u32)roots.size(), HSPF_RUN_NET_NEXTHOPS, 1, d.data(), h.data(), f.data(), pr.data(), m.data(),
                   t.mask_words, nn.data(), np.data(), wk.data()) == 0, "oracle failed");
      CHECK(d == t.dist && h == t.hops && m == t.mask, "cost-patched graph differs from the oracle");
      ++cost_patched;
    }
    std::vector<uint8_t> leaf(g.n); size_t lb = 0;
    CHECK(hspf_graph_export(eng.raw(), G.raw(), HSPF_GX_LEAF, leaf.data(), leaf.size(), &lb) == HSPF_OK && lb == g.n, "HSPF_GX_LEAF");
    hspf_rows bad{1, nullptr, nullptr, nullptr, nullptr, nullptr};
    CHECK(hspf_graph_patch(eng.raw(), G.raw(), &bad) == HSPF_E_INVAL, "NULL arrays in a patch must be HSPF_E_INVAL");
  }
  // several GPUs through the C ABI: three contexts on device 0 (the list may repeat an ordinal), tables sized for ALL
  // roots on every "device", in-place gather by device-to-device copies; synchronous and asynchronous
  int sharded = 0;
  {
    Lsdb g = make(260, 14, 33);
    std::vector<u32> roots; for (u32 r = 14; r < 14 + 200; ++r) roots.push_back(r);      // 4 batches: 2 + 1 + 1
    hspf::Graph G = eng.upload(g.row_ptr, g.col, g.metric, g.vflags, 0xFE000000u);
    hspf::Tables t = eng.run(G, roots, 0);
    const u32 R = (u32)roots.size(), n = g.n, W = t.mask_words;
    const int devs[3] = {0, 0, 0};
    hspf_multi_config cfg{3, devs, 3, 0, nullptr};
    hspf_multi *m = nullptr;
    CHECK(hspf_multi_init(&cfg, &m) == HSPF_OK && hspf_multi_n_local(m) == 3, "hspf_multi_init");
    hspf_csr csr{g.n, (u32)g.col.size(), g.row_ptr.data(), g.col.data(), g.metric.data(), g.vflags.data(), 0xFE000000u};
    hspf_multi_graph *mg = nullptr;
    CHECK(hspf_multi_graph_upload(m, &csr, &mg) == HSPF_OK, "hspf_multi_graph_upload");
    u32 w2 = 0;
    CHECK(hspf_multi_mask_words(m, mg, roots.data(), R, &w2) == HSPF_OK && w2 == W, "hspf_multi_mask_words");
    u32 b, e, covered = 0;
    for (u32 r = 0; r < 3; ++r) { hspf_shard_bounds(R, 3, r, &b, &e); CHECK(b == covered && b % 64 == 0, "shard bounds"); covered = e; }
    CHECK(covered == R, "shard bounds cover");
    hspf_result all[3];
    for (int i = 0; i < 3; ++i) {
      u32 *dd; uint16_t *dh, *df; uint64_t *dm;
      hipMalloc(&dd, (size_t)R * n * 4); hipMalloc(&dh, (size_t)R * n * 2); hipMalloc(&df, (size_t)R * n * 2); hipMalloc(&dm, (size_t)R * n * 8 * W);
      all[i] = hspf_result{dd, dh, df, dm, W, nullptr};
    }
    for (u32 mode : {(u32)(HSPF_GATHER_DIST | HSPF_GATHER_HOPS | HSPF_GATHER_FLAGS | HSPF_GATHER_MASK),
                     (u32)(HSPF_GATHER_DIST | HSPF_GATHER_HOPS | HSPF_GATHER_FLAGS | HSPF_GATHER_MASK | HSPF_GATHER_ASYNC)}) {
      for (int i = 0; i < 3; ++i) { hipMemset(all[i].dist, 0, (size_t)R * n * 4); hipMemset(all[i].first_hop_mask, 0, (size_t)R * n * 8 * W); }
      hipDeviceSynchronize();
      CHECK(hspf_multi_run(m, mg, roots.data(), R, 0, all, mode) == HSPF_OK, hspf_multi_last_error(m));
      CHECK(hspf_multi_wait(m) == HSPF_OK, "hspf_multi_wait");
      for (int i = 0; i < 3; ++i) {
        std::vector<u32> d((size_t)R * n); std::vector<uint16_t> h((size_t)R * n), f((size_t)R * n); std::vector<uint64_t> mk((size_t)R * n * W);
        hipMemcpy(d.data(), all[i].dist, d.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(h.data(), all[i].hops, h.size() * 2, hipMemcpyDeviceToHost);
        hipMemcpy(f.data(), all[i].vflags_out, f.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(mk.data(), all[i].first_hop_mask, mk.size() * 8, hipMemcpyDeviceToHost);
        CHECK(d == t.dist && h == t.hops && mk == t.mask, "sharded run differs from the unsharded one");
        for (size_t k = 0; k < f.size(); ++k) CHECK((f[k] & 1) == (t.flags[k] & 1), "sharded flags");
        ++sharded;
      }
    }
    for (int i = 0; i < 3; ++i) { hipFree(all[i].dist); hipFree(all[i].hops); hipFree(all[i].vflags_out); hipFree(all[i].first_hop_mask); }
    // areas first, then roots: 3 areas on 2 ranks, every (area, batch) unit exactly once, loads within one batch
    const u32 rpa[3] = {130, 64, 200};
    hspf_area_slice sl[8];
    const u32 ns = hspf_plan_areas(3, rpa, 2, sl, 8);
    u32 load[2] = {0, 0}, seen[3] = {0, 0, 0};
    for (u32 k = 0; k < ns; ++k) { CHECK(sl[k].root_begin == seen[sl[k].area], "area plan order"); seen[sl[k].area] = sl[k].root_end; load[sl[k].rank] += (sl[k].root_end - sl[k].root_begin + 63) / 64; }
    CHECK(seen[0] == 130 && seen[1] == 64 && seen[2] == 200 && load[0] + load[1] == 8 && (load[0] > load[1] ? load[0] - load[1] : load[1] - load[0]) <= 1, "area plan");
    hspf_multi_graph_free(m, mg);
    hspf_multi_shutdown(m);
  }
  // a LAN of 3 000 routers on a 4 000-router graph: the graph is built from sorted keys (hub mode) and the pseudonode's
  // row is evaluated in slices (giant row); roots off the LAN (a member would need 3 000+ first-hop slots)
  int big_lan = 0;
  {
    rng_state = 0xABCDEF12345ull;
    const u32 n_rtr = 4000, n = n_rtr + 1, lan = 3000;
    std::vector<std::vector<std::pair<u32, u32>>> rows(n);
    for (u32 i = 0; i < n_rtr * 2; ++i) {
      const u32 u = 1 + rnd(n_rtr), v = 1 + rnd(n_rtr);
      if (u == v) continue;
      rows[u].push_back({v, 1 + rnd(9)}); rows[v].push_back({u, 1 + rnd(9)});
    }
    for (u32 r = 1; r <= lan; ++r) { rows[r].push_back({0, 1 + rnd(9)}); rows[0].push_back({r, 0}); }
    Lsdb g; g.n = n; g.row_ptr.assign(n + 1, 0); g.vflags.assign(n, 0); g.vflags[0] = HSPF_VF_NETWORK;
    for (u32 u = 0; u < n; ++u) {
      for (auto &e : rows[u]) { g.col.push_back(e.first); g.metric.push_back(e.second); }
      g.row_ptr[u + 1] = (u32)g.col.size();
      if (u && rnd(100) < 3) g.vflags[u] |= HSPF_VF_NO_TRANSIT;
    }
    hspf::Graph G = eng.upload(g.row_ptr, g.col, g.metric, g.vflags, 0xFE000000u);
    u32 mode = 7; size_t got = 0;
    CHECK(hspf_graph_export(eng.raw(), G.raw(), HSPF_GX_BUILD_MODE, &mode, 4, &got) == HSPF_OK && got == 4 && mode == 1, "hub-mode build expected");
    std::vector<u32> roots;
    for (u32 r = lan + 1; r < lan + 1 + 100; ++r) roots.push_back(r);         // off the LAN: one mask word
    for (u32 flags : {0u, (u32)(HSPF_RUN_NET_NEXTHOPS | HSPF_RUN_IGNORE_OVERLOAD)}) {
      hspf::Tables t = eng.run(G, roots, flags);
      CHECK(t.mask_words == 1, "roots off the LAN need one mask word");
      const size_t rn = (size_t)roots.size() * n;
      std::vector<u32> d(rn), pr(rn), nn(rn), np(rn); std::vector<uint16_t> h(rn), f(rn); std::vector<uint64_t> mk(rn), wk(roots.size());
      CHECK(oracle(n, (u32)g.col.size(), g.row_ptr.data(), g.col.data(), g.metric.data(), g.vflags.data(), 0xFE000000u, roots.data(),
                   (u32)roots.size(), flags, /*variant HEAP*/ 2, d.data(), h.data(), f.data(), pr.data(), mk.data(), 1, nn.data(),
                   np.data(), wk.data()) == 0, "oracle failed");
      CHECK(d == t.dist && h == t.hops && mk == t.mask, "big LAN: run differs from the oracle");
      ++big_lan;
    }
  }
  // ABI 7: packed results through the C ABI — hspf_run_packed (page-locked and pageable destinations), hspf_run_packed_async
  // with three tickets in flight, hspf_run_packed_device; every word decoded with the header's inline helpers against the
  // oracle; HSPF_E_NO_PACKED for roots on a 40-router LAN
  int packed_runs = 0;
  {
    rng_state = 0x5EED5EED5EEDull;
    Lsdb g = make(900, 0, 71);
    hspf::Graph G = eng.upload(g.row_ptr, g.col, g.metric, g.vflags, 0xFE000000u);
    std::vector<std::vector<u32>> sets;
    for (u32 base : {0u, 100u, 300u}) { std::vector<u32> r; for (u32 k = 0; k < 64 + base / 100; ++k) r.push_back(base + k); sets.push_back(r); }
    auto check_words = [&](const std::vector<u32> &roots, const hspf_packed_layout &ly, const void *words, const uint8_t *status) -> int {
      const size_t rn = (size_t)roots.size() * g.n;
      std::vector<u32> d(rn), pr(rn), nn(rn), np(rn); std::vector<uint16_t> h(rn), f(rn); std::vector<uint64_t> m(rn), wk(roots.size());
      CHECK(oracle(g.n, (u32)g.col.size(), g.row_ptr.data(), g.col.data(), g.metric.data(), g.vflags.data(), 0xFE000000u, roots.data(), (u32)roots.size(), 0, 2,
                   d.data(), h.data(), f.data(), pr.data(), m.data(), 1, nn.data(), np.data(), wk.data()) == 0, "oracle failed");
      CHECK(ly.word_bytes == 4 || ly.word_bytes == 8, "word_bytes");
      for (size_t i = 0; i < rn; ++i) {
        const uint64_t w = hspf_packed_word(&ly, words, i);
        CHECK((hspf_packed_in_spt(&ly, w) != 0) == ((f[i] & 1) != 0) && hspf_packed_dist(&ly, w) == d[i] && hspf_packed_hops(&ly, w) == h[i] && hspf_packed_mask(&ly, w) == m[i],
              "packed word differs from the oracle");
      }
      for (size_t r = 0; r < roots.size(); ++r) CHECK((status[r] & ~HSPF_ROOT_EXACT) == 0, "root status");
      ++packed_runs;
      return 0;
    };
    const size_t cap = (size_t)8 * 70 * g.n;
    void *pin[3] = {nullptr, nullptr, nullptr};
    for (auto &p : pin) CHECK(hspf_host_alloc(eng.raw(), cap, &p) == HSPF_OK && p, "hspf_host_alloc");
    std::vector<uint8_t> pageable(cap), status(70);
    for (auto &roots : sets) {
      hspf_packed_layout ly{};
      CHECK(hspf_run_packed(eng.raw(), G.raw(), roots.data(), (u32)roots.size(), 0, pin[0], cap, &ly, status.data()) == HSPF_OK, hspf_last_error(eng.raw()));
      if (check_words(roots, ly, pin[0], status.data())) return 1;
      CHECK(hspf_run_packed(eng.raw(), G.raw(), roots.data(), (u32)roots.size(), 0, pageable.data(), cap, &ly, status.data()) == HSPF_OK, hspf_last_error(eng.raw()));
      if (check_words(roots, ly, pageable.data(), status.data())) return 1;
    }
    uint64_t tk[3]; std::vector<uint8_t> st3[3];
    for (int i = 0; i < 3; ++i) { st3[i].assign(70, 0); CHECK(hspf_run_packed_async(eng.raw(), G.raw(), sets[i].data(), (u32)sets[i].size(), 0, pin[i], cap, st3[i].data(), &tk[i]) == HSPF_OK, "hspf_run_packed_async"); }
    for (int i = 0; i < 3; ++i) {
      hspf_packed_layout ly{}; hspf_stats st{};
      CHECK(hspf_wait_packed(eng.raw(), tk[i], &ly, &st) == HSPF_OK && st.n_roots == sets[i].size(), hspf_last_error(eng.raw()));
      if (check_words(sets[i], ly, pin[i], st3[i].data())) return 1;
    }
    {
      void *dw = nullptr; hipMalloc(&dw, cap);
      hspf_packed_layout ly{};
      CHECK(hspf_run_packed_device(eng.raw(), G.raw(), sets[2].data(), (u32)sets[2].size(), 0, dw, cap, &ly, status.data()) == HSPF_OK, hspf_last_error(eng.raw()));
      hipMemcpy(pageable.data(), dw, (size_t)ly.word_bytes * sets[2].size() * g.n, hipMemcpyDeviceToHost);
      if (check_words(sets[2], ly, pageable.data(), status.data())) return 1;
      hipFree(dw);
    }
    hspf_packed_layout ly{};
    CHECK(hspf_run_packed(eng.raw(), G.raw(), sets[0].data(), 64, HSPF_RUN_POP_RANK, pin[0], cap, &ly, nullptr) == HSPF_E_INVAL, "pop rank with packed words must be HSPF_E_INVAL");
    CHECK(hspf_run_packed(eng.raw(), G.raw(), sets[0].data(), 64, 0, pin[0], 1000, &ly, nullptr) == HSPF_E_INVAL, "a too small buffer must be HSPF_E_INVAL");
    // a LAN of 40 routers: its members have 40+ first-hop slots
    Lsdb l; l.n = 60; l.vflags.assign(60, 0); l.vflags[0] = HSPF_VF_NETWORK;
    std::vector<std::vector<std::pair<u32, u32>>> rows(60);
    for (u32 r = 1; r <= 40; ++r) { rows[r].push_back({0, 10}); rows[0].push_back({r, 0}); }
    for (u32 r = 41; r < 60; ++r) { rows[r].push_back({r - 1, 3}); rows[r - 1].push_back({r, 3}); }
    l.row_ptr.assign(61, 0);
    for (u32 u = 0; u < 60; ++u) { for (auto &e : rows[u]) { l.col.push_back(e.first); l.metric.push_back(e.second); } l.row_ptr[u + 1] = (u32)l.col.size(); }
    hspf::Graph GL = eng.upload(l.row_ptr, l.col, l.metric, l.vflags, 0xFE000000u);
    const u32 on_lan[2] = {1, 50}, off_lan[2] = {50, 55};
    CHECK(hspf_run_packed(eng.raw(), GL.raw(), on_lan, 2, 0, pin[0], cap, &ly, nullptr) == HSPF_E_NO_PACKED, "more than 24 slots must be HSPF_E_NO_PACKED");
    CHECK(hspf_run_packed(eng.raw(), GL.raw(), off_lan, 2, 0, pin[0], cap, &ly, status.data()) == HSPF_OK, hspf_last_error(eng.raw()));
    hspf::PackedTables pt = eng.run_packed(GL, std::vector<u32>{50, 55});
    hspf::Tables tt = eng.run(GL, std::vector<u32>{50, 55});
    for (u32 r = 0; r < 2; ++r) for (u32 v = 0; v < 60; ++v)
      CHECK(pt.dist(r, v) == tt.dist[r * 60 + v] && pt.hops(r, v) == tt.hops[r * 60 + v] && pt.mask(r, v) == tt.mask[r * 60 + v] && pt.in_spt(r, v) == ((tt.flags[r * 60 + v] & 1) != 0), "PackedTables accessors");
    for (auto &p : pin) hspf_host_free(eng.raw(), p);
  }
  std::printf("capi_parity: %d packed-result runs (pinned, pageable, in flight, device) bit-exact after decode\n", packed_runs);
  std::printf("capi_parity: %d graph runs bit-exact, %d patched generations + %d in-place cost patches bit-exact, error contract ok, device route derivation ok, %d packed record stream(s) ok, %d sharded tables identical to the unsharded run, %d runs on a 3 000-router LAN (hub-mode build, giant row) bit-exact, %d asynchronous runs (in flight on the lanes) identical to the synchronous ones\n", checked, patched, cost_patched, packed, sharded, big_lan, async_checked);
  return 0;
}
