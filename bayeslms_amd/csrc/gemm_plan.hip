// Launch planning of the fp32 MFMA GEMM family (host code): tile shape x split-K count for one blm_gemm call.
//
// The reference leaves this choice to the vendor BLAS behind F.linear (model.py:1127-1129 and every other call site);
// here it is explicit and inspectable (blm_gemm_plan_query runs without a GPU):
//   1. a process-wide override (blm_gemm_plan_override / BLM_GEMM_TILE, BLM_GEMM_SPLITK): tuning tools only;
//   2. the plan table: exact (layout, M, N, K, epilogue, accumulate) keys with the plan that was fastest INSIDE the
//      step it belongs to (tools/gemm_tune.py times every candidate in situ -- stand-alone timings run colder and rank
//      the tiles differently -- and writes gemm_plans.inc; blm_gemm_plan_set adds entries at run time);
//   3. the cost model below for every other shape.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "blm_host.h"
#include "gemm_plan.h"

namespace blm {

namespace {

struct Entry { int op, M, N, K, epi, acc, tile, splits; };

const Entry kBuiltin[] = {
#include "gemm_plans.inc"
    {-1, 0, 0, 0, 0, 0, 0, 0}};

// Plans for launches that run BESIDE a collective's channel workgroups (blm_gemm_plan_comm_window): measured stand-alone
// next to a resident stand-in with RCCL's channel-kernel footprint (tools/gemm_tune_comm.py writes gemm_plans_comm.inc).
// Listed: shapes whose whole-chip plan loses more than 5 % to another plan there; every other launch keeps its plan.
const Entry kComm[] = {
#include "gemm_plans_comm.inc"
    {-1, 0, 0, 0, 0, 0, 0, 0}};

std::mutex g_mu;
std::vector<Entry> g_runtime;   // blm_gemm_plan_set
bool g_builtin_on = true;       // blm_gemm_plan_clear switches the built-in table off as well (model only)
int g_force_tile = -1, g_force_splits = 0;  // tile -1: environment not read yet
// Compute units the plans may count on (blm_gemm_plan_set_cus).  Every plan-table entry was measured with the whole chip to
// itself -- the roofline launch is exactly ONE round of 256 one-per-CU workgroups -- so the table applies at kChipCUs only;
// with fewer CUs (a collective's channel workgroups resident beside the GEMMs) the cost model plans for that many.
int g_cus = kChipCUs;
// Modelled GEMM time (us) that is still expected to run beside gradient buckets in flight (blm_gemm_plan_comm_window): the
// host enqueues far ahead of the device, so "while a bucket is in flight" is kept in DEVICE time -- every bucket adds its
// expected time on the links, every planned launch takes its own modelled time off.
double g_window_us = 0.0;
std::vector<Entry> g_comm_runtime;  // blm_gemm_plan_set_comm
bool g_comm_builtin_on = true;

// choose_plan runs in front of EVERY blm_gemm launch (incl. the per-time-step products of the step-wise recurrent paths):
// the plan of a key is computed once and kept until something that can change it happens (set / clear / override / set_cus).
struct KeyHash {
  size_t operator()(const PlanKey& k) const {
    size_t h = 1469598103934665603ull;
    for (int v : {k.op, k.M, k.N, k.K, k.epi, k.acc, k.can_split, k.fast}) { h ^= (size_t)(unsigned)v; h *= 1099511628211ull; }
    return h;
  }
};
struct KeyEq {
  bool operator()(const PlanKey& a, const PlanKey& b) const {
    return a.op == b.op && a.M == b.M && a.N == b.N && a.K == b.K && a.epi == b.epi && a.acc == b.acc && a.can_split == b.can_split &&
           a.fast == b.fast;
  }
};
// [cus, or -1 inside a comm window][key]: both switches toggle several times per training step
std::unordered_map<int, std::unordered_map<PlanKey, Plan, KeyHash, KeyEq>> g_memo;

// ---- cost model -------------------------------------------------------------------------------------------------
// A workgroup is 4 waves, one per SIMD; per K tile of 32 a wave issues wtm*wtn*16 v_mfma_f32_32x32x2_f32 of 64 cycles.
// occ workgroups share a CU (LDS: 32 / 48 / 48 / 64 KB of 160), so a full round of 256*occ workgroups costs
// occ * kt * 1024*wtm*wtn cycles of matrix issue per SIMD, stretched by the tile's efficiency e(o) = einf * o / (o + a)
// at o co-resident workgroups (one wave per SIMD hides nothing, five hide almost everything), plus a fixed cost per
// co-resident workgroup (prologue, epilogue, C store); the last, partly filled round runs o = ceil(rest / 256) per CU.
// Operand / output streaming enters as a soft maximum (near that bound the plan still matters), split-K adds the atomic
// read-modify-write of C once per slice and a zeroing pass when C is not accumulated into.  The constants are a
// least-squares fit (tools/gemm_fit.py) to the stand-alone sweep of tools/gemm_tune.py --grid on one MI355X -- 2160
// shapes x 5 tiles x up to 8 slice counts = 61,944 timings, median |log error| 4 %; the model's PICK reaches 98.5 % of
// the best candidate's rate on average, >= 0.78 on every shape above 30 us (profiles/r03_gemm_model_fit.txt).
struct TileModel { int tile, wtm, wtn, occ; };
// 28 = the 128 x 128 tile on eight waves: the output tile, LDS footprint and matrix time per SIMD of tile 22 (two waves of
// 64 x 32 per SIMD instead of one of 64 x 64), its own efficiency constants
constexpr int kNTiles = 5;
const TileModel kTiles[kNTiles] = {{11, 1, 1, 5}, {12, 1, 2, 3}, {21, 2, 1, 3}, {22, 2, 2, 2}, {28, 2, 2, 2}};
struct ModelK {
  double cyc_per_us;        // nominal matrix-pipe clock; einf absorbs the clock under load
  double einf[3][kNTiles];  // [layout][tile]: efficiency of the MFMA issue with many co-resident workgroups
  double a[kNTiles];        // e(o) = einf * o / (o + a)
  double t0[kNTiles];       // fixed cost of the first round (us)
  double t0r[kNTiles];      // ... of every further round
  double launch_us;
  double atomic_bpus;       // bytes of C per microsecond through float atomics
  double memset_bpus, memset_us;
  double hbm_bpus;          // streaming rate of A + B + C
  double t0o[kNTiles];      // fixed cost per co-resident workgroup of a round (us)
};
ModelK g_model = {
    2400.0,
    {{0.9594, 1.0617, 1.0602, 1.0452, 0.9315}, {0.9580, 1.0628, 1.0633, 1.0481, 0.9240}, {0.9268, 1.0219, 1.0221, 1.0193, 0.9714}},
    {0.2471, 0.3049, 0.2937, 0.1959, 0.0224},
    {0.540, 0.652, 0.545, 2.132, 2.989},
    {0.000, 0.000, 0.000, 0.000, 0.000},
    3.458,
    3.567e+06,
    1.015e+07, 2.162,
    4.31e+06,
    {0.269, 1.342, 1.213, 2.697, 1.505},
};

// S >= 1: every tile in S slices.  S <= -2: tail slicing -- the whole rounds of workgroup slots compute their tiles in one
// piece (no atomics), only the tiles of the last, partly filled round are cut |S| ways (gemm_f32_mfma.h launch_cfg).
double model_us(const PlanKey& k, const TileModel& t, int S) {
  const ModelK& m = g_model;
  const int ti = (int)(&t - kTiles);
  const long BM = 64 * t.wtm, BN = 64 * t.wtn;
  const long tiles = ((k.M + BM - 1) / BM) * ((k.N + BN - 1) / BN);
  const long slots = (long)g_cus * t.occ;
  auto eff = [&](double o) { return m.einf[k.op][ti] * o / (o + m.a[ti]); };
  auto tk_of = [&](long kper) { return (double)((kper + 31) / 32) * 1024.0 * t.wtm * t.wtn / m.cyc_per_us; };  // matrix issue time of one workgroup on its SIMDs
  bool tail = S < -1;
  long full, rem;       // whole rounds, workgroups of the last round
  double tk_full, tk_rem, sliced_bytes;
  if (tail && tiles / slots > 0 && tiles % slots != 0) {
    const long St = -S, rt = tiles % slots;
    full = tiles / slots; rem = rt * St;
    tk_full = tk_of(k.K); tk_rem = tk_of((k.K + St - 1) / St);
    sliced_bytes = (double)St * (double)rt * BM * BN * 4.0;
    if (rem > slots) return 1e30;  // more slices than one round holds: not a tail plan
  } else {
    if (tail) { S = tiles % slots == 0 ? 1 : -S; tail = false; }  // no whole round (or nothing left over): the uniform form
    const long G = tiles * S;
    full = G / slots; rem = G - full * slots;
    tk_full = tk_rem = tk_of((k.K + S - 1) / S);
    sliced_bytes = S > 1 ? (double)S * (double)k.M * k.N * 4.0 : 0.0;
  }
  const double t_full = (double)t.occ * tk_full / eff((double)t.occ) + m.t0o[ti] * t.occ;
  double us = m.launch_us;
  if (full > 0) us += m.t0[ti] + t_full + (double)(full - 1) * (m.t0r[ti] + t_full);
  if (rem > 0) {
    const double o = (double)((rem + g_cus - 1) / g_cus);
    us += (full > 0 ? m.t0r[ti] : m.t0[ti]) + o * tk_rem / eff(o) + m.t0o[ti] * o;
  }
  const double bytes = 4.0 * ((double)k.M * k.K + (double)k.N * k.K + (double)k.M * k.N * (k.acc ? 2.0 : 1.0));
  const double mem = m.launch_us + bytes / m.hbm_bpus;
  us = std::cbrt(us * us * us + mem * mem * mem);
  if (sliced_bytes > 0.0) {
    us += sliced_bytes / m.atomic_bpus;
    if (!k.acc) us += m.memset_us + (double)k.M * k.N * 4.0 / m.memset_bpus;
  }
  return us;
}

Plan model_plan(const PlanKey& k) {
  Plan best{11, 1, 0};
  double bt = 1e30;
  for (const TileModel& t : kTiles) {
    if (!k.fast && t.tile != 11) continue;
    if (t.tile == 28 && k.K % 32 != 0) continue;  // the eight-wave tile has no K-tail path
    for (int S = 1; S <= 16; ++S) {
      if (S > 1 && (!k.can_split || k.K / S < 128)) break;
      if (S == 5 || S == 7 || (S > 8 && S != 12 && S != 16)) continue;  // the slice counts the sweep measured
      double us = model_us(k, t, S) * (1.0 + 0.002 * S);  // among equals, the fewest slices
      // K slices through the bias epilogue (legal since late round 3, not in the sweep the constants were fitted to) cost
      // ~10 us more than the fit predicts at the small sizes where it matters (tools/gemm_bias_pick_check.py: the model alone
      // picked 11/4 for NT 2560 x 512 x 1024 + bias, 44.8 us against 35.9 us unsliced; with this term the mean of best / chosen over 63 shapes goes
      // 0.957 -> 0.98)
      if (S > 1 && k.epi == BLM_EPI_BIAS) {
        const long tl = ((k.M + 64L * t.wtm - 1) / (64L * t.wtm)) * ((k.N + 64L * t.wtn - 1) / (64L * t.wtn));
        us += tl >= 128 ? 10.0 : 4.0;  // less where the unsliced grid leaves most of the chip empty
      }
      if (us < bt) { bt = us; best.tile = t.tile; best.splits = S; }
    }
    // (tail-sliced plans -- only the tiles beyond the last whole round are cut -- are scored by model_us but not proposed here:
    // the sweep the constants were fitted to has none; the plan table carries the measured ones)
  }
  return best;
}


const Entry* find(const PlanKey& k) {
  for (auto it = g_runtime.rbegin(); it != g_runtime.rend(); ++it)
    if (it->op == k.op && it->M == k.M && it->N == k.N && it->K == k.K && it->epi == k.epi && it->acc == k.acc) return &*it;
  if (g_builtin_on)
    for (const Entry* e = kBuiltin; e->op >= 0; ++e)
      if (e->op == k.op && e->M == k.M && e->N == k.N && e->K == k.K && e->epi == k.epi && e->acc == k.acc) return e;
  return nullptr;
}

const Entry* find_comm(const PlanKey& k) {
  for (auto it = g_comm_runtime.rbegin(); it != g_comm_runtime.rend(); ++it)
    if (it->op == k.op && it->M == k.M && it->N == k.N && it->K == k.K && it->epi == k.epi && it->acc == k.acc) return &*it;
  if (g_comm_builtin_on)
    for (const Entry* e = kComm; e->op >= 0; ++e)
      if (e->op == k.op && e->M == k.M && e->N == k.N && e->K == k.K && e->epi == k.epi && e->acc == k.acc) return e;
  return nullptr;
}

// 28 = 128 x 128 on eight waves (gemm_f32_mfma.h): needs the aligned fast path and whole K tiles, else the 4-wave 128 x 128 tile runs
bool valid_tile(int t) { return t == 11 || t == 12 || t == 21 || t == 22 || t == 28; }

void read_env() {
  if (g_force_tile >= 0) return;
  const char* e = getenv("BLM_GEMM_TILE");
  g_force_tile = e ? atoi(e) : 0;
  e = getenv("BLM_GEMM_SPLITK");
  g_force_splits = e ? atoi(e) : 0;
  // BLM_GEMM_PLAN_SET="op,M,N,K,epilogue,accumulate,tile,splits[;...]": run-time table entries from the environment
  // (profiling one launch under another plan without touching the others: tools/profile_round.sh)
  e = getenv("BLM_GEMM_PLAN_SET");
  while (e && *e) {
    int v[8], n = 0;
    char* end = nullptr;
    while (n < 8) {
      v[n++] = (int)strtol(e, &end, 10);
      if (end == e) { n = 0; break; }
      e = end;
      if (*e == ',') ++e; else break;
    }
    if (n == 8 && v[0] >= 0 && v[0] <= 2 && valid_tile(v[6]) && v[7] >= -64 && v[7] <= 64 && v[7] != 0 && v[7] != -1)
      g_runtime.push_back(Entry{v[0], v[1], v[2], v[3], v[4], v[5] ? 1 : 0, v[6], v[7]});
    while (*e && *e != ';') ++e;
    if (*e == ';') ++e;
  }
}


}  // namespace

double plan_model_us(const PlanKey& k, int tile, int splits) {
  std::lock_guard<std::mutex> lk(g_mu);  // model_us reads g_cus (found by the ThreadSanitizer build: a query beside blm_gemm_plan_set_cus)
  for (const TileModel& t : kTiles)
    if (t.tile == tile) return model_us(k, t, (splits == 0 || splits == -1) ? 1 : splits);
  return -1.0;
}

int plan_cus() {
  std::lock_guard<std::mutex> lk(g_mu);
  return g_cus;
}

Plan choose_plan(const PlanKey& k, bool consume) {
  std::lock_guard<std::mutex> lk(g_mu);
  read_env();
  const bool under_comm = g_window_us > 0.0 && g_cus == kChipCUs;
  auto& memo = g_memo[under_comm ? -1 : g_cus];
  auto hit = memo.find(k);
  if (hit != memo.end()) {
    if (under_comm && consume) g_window_us -= hit->second.us;
    return hit->second;
  }
  Plan p{11, 1, 0, g_cus, 0.f};
  // table entries were measured on the whole chip: they apply there only
  const Entry* e = g_cus != kChipCUs ? nullptr : find(k);  // blm_gemm_plan_clear(0) switches the table off (cost model only)
  // a plan measured WITH K slices says nothing about the tile to run WITHOUT them (deterministic mode, or a launch whose C cannot
  // take atomics): the cost model then picks among the unsliced candidates
  if (e && !k.can_split && (e->splits > 1 || e->splits < -1)) e = nullptr;
  if (e) { p.tile = e->tile; p.splits = e->splits; p.source = 1; }
  else { p = model_plan(k); p.cus = g_cus; }
  // beside a collective: the plan measured there, where one exists (everything else keeps its whole-chip plan -- in situ the
  // cost model's plans for a narrowed chip lose to the measured table whenever no channel workgroup is resident)
  if (under_comm)
    if (const Entry* c = find_comm(k))
      if (k.can_split || (c->splits >= -1 && c->splits <= 1)) { p.tile = c->tile; p.splits = c->splits; p.source = 3; }
  if (g_force_tile > 0 && valid_tile(g_force_tile)) { p.tile = g_force_tile; p.source = 2; }
  if (g_force_splits != 0) { p.splits = g_force_splits; p.source = 2; }
  // legality, whatever the source said (splits <= -2: tail slicing, |splits| ways)
  if (!k.fast) p.tile = 11;
  if (p.tile == 28 && k.K % 32 != 0) p.tile = 22;
  if (!k.can_split || p.splits == 0 || p.splits == -1) p.splits = 1;
  int n = p.splits < 0 ? -p.splits : p.splits;
  while (n > 1 && k.K / n < 32) --n;
  p.splits = n < 2 ? 1 : (p.splits < 0 ? -n : n);
  if (p.splits < 0) {  // canonical form: a tail plan without a whole round is the uniform plan, one without a remainder is unsliced
    for (const TileModel& t : kTiles)
      if (t.tile == p.tile) {
        const long tiles = ((k.M + 64L * t.wtm - 1) / (64L * t.wtm)) * ((k.N + 64L * t.wtn - 1) / (64L * t.wtn)), slots = (long)g_cus * t.occ;
        if (tiles < slots) p.splits = -p.splits;
        else if (tiles % slots == 0) p.splits = 1;
      }
  }
  for (const TileModel& t : kTiles)
    if (t.tile == p.tile) p.us = (float)model_us(k, t, p.splits);
  if (memo.size() > 4096) memo.clear();  // step-wise paths with ever-changing shapes: bounded
  memo.emplace(k, p);
  if (under_comm && consume) g_window_us -= p.us;
  return p;
}

}  // namespace blm

using namespace blm;

static int key_of(const blm_gemm_args* a, PlanKey* k) {
  if (!a) return blm_fail(BLM_ERR_INVALID, "blm_gemm_plan_query: null args");
  if (a->abi_version != BLM_ABI_VERSION) return blm_fail(BLM_ERR_ABI, "blm_gemm_plan_query: abi_version mismatch");
  if (a->op < BLM_GEMM_NT || a->op > BLM_GEMM_TN || a->M <= 0 || a->N <= 0 || a->K < 0)
    return blm_fail(BLM_ERR_INVALID, "blm_gemm_plan_query: bad op or shape");
  *k = plan_key(a);
  return BLM_OK;
}

blm::PlanKey blm::plan_key(const blm_gemm_args* a) {
  PlanKey k{};
  k.op = a->op; k.M = a->M; k.N = a->N; k.K = a->K; k.epi = a->epilogue;
  k.acc = (a->flags & BLM_GEMM_ACCUMULATE) ? 1 : 0;
  const bool samp = a->var_b.lgstd != nullptr;
  // split-K: partial sums meet in C through float atomics -- only for the plain and bias epilogues (C zeroed first unless
  // accumulating, which needs a dense C; the bias rides on the first slice) and for the Bayesian wgrad epilogue, which is
  // linear in dW (KL terms from the first slice only) but has no zeroing pass for its second output
  const bool lin = a->epilogue == BLM_EPI_NONE || a->epilogue == BLM_EPI_BIAS;
  k.can_split = !samp && ((lin && (k.acc || a->ldc == a->N)) || (a->epilogue == BLM_EPI_BAYES_WGRAD && k.acc));
  // deterministic mode (blm_set_option("deterministic", 1)): partial sums never meet through float atomics -- every plan, whatever
  // its source (table, override, model), is legalised to one K slice, tail slicing included (choose_plan: !can_split -> splits = 1)
  if (option(OPT_DETERMINISTIC)) k.can_split = 0;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const int ac = a->op == BLM_GEMM_TN ? a->M : a->K, bc = a->op == BLM_GEMM_NT ? a->K : a->N;
  k.fast = al16(a->A) && al16(a->B) && a->lda % 4 == 0 && a->ldb % 4 == 0 && ac % 4 == 0 && bc % 4 == 0 && ac >= 4 && bc >= 4;
  const long arows = a->op == BLM_GEMM_TN ? a->K : a->M, brows = a->op == BLM_GEMM_NT ? a->N : a->K;
  if (arows * (long)a->lda * 4 >= (1L << 32) || brows * (long)a->ldb * 4 >= (1L << 32)) k.fast = 0;
  return k;
}

extern "C" int blm_gemm_plan_query(const blm_gemm_args* a, blm_gemm_plan* out) {
  PlanKey k;
  if (int rc = key_of(a, &k)) return rc;
  if (!out) return blm_fail(BLM_ERR_INVALID, "blm_gemm_plan_query: null out");
  const Plan p = choose_plan(k, /*consume=*/false);  // a query is not a launch: an open comm window keeps its time
  out->tile = p.tile; out->splits = p.splits; out->source = p.source;
  out->model_us = (float)plan_model_us(k, p.tile, p.splits);
  return BLM_OK;
}

extern "C" int blm_gemm_plan_launch(const blm_gemm_args* a, blm_gemm_plan* out) {
  PlanKey k;
  if (int rc = key_of(a, &k)) return rc;
  if (!out) return blm_fail(BLM_ERR_INVALID, "blm_gemm_plan_launch: null out");
  const Plan p = choose_plan(k, /*consume=*/true);
  out->tile = p.tile; out->splits = p.splits; out->source = p.source;
  out->model_us = (float)plan_model_us(k, p.tile, p.splits);
  return BLM_OK;
}

extern "C" int blm_gemm_plan_model_us(const blm_gemm_args* a, int tile, int splits, float* us) {
  PlanKey k;
  if (int rc = key_of(a, &k)) return rc;
  const double v = plan_model_us(k, tile, splits);
  if (v < 0 || !us) return blm_fail(BLM_ERR_INVALID, "blm_gemm_plan_model_us: unknown tile %d", tile);
  *us = (float)v;
  return BLM_OK;
}

extern "C" int blm_gemm_plan_override(int tile, int splits) {
  if (tile != 0 && !valid_tile(tile)) return blm_fail(BLM_ERR_INVALID, "blm_gemm_plan_override: tile must be 0, 11, 12, 21, 22 or 28");
  if (splits < -64 || splits > 64 || splits == -1) return blm_fail(BLM_ERR_INVALID, "blm_gemm_plan_override: splits out of range");
  std::lock_guard<std::mutex> lk(g_mu);
  g_force_tile = tile; g_force_splits = splits;
  g_memo.clear();
  return BLM_OK;
}

extern "C" int blm_gemm_plan_set(int op, int M, int N, int K, int epilogue, int accumulate, int tile, int splits) {
  if (op < BLM_GEMM_NT || op > BLM_GEMM_TN || !valid_tile(tile) || splits < -64 || splits > 64 || splits == 0 || splits == -1)
    return blm_fail(BLM_ERR_INVALID, "blm_gemm_plan_set: bad op, tile or split count");
  std::lock_guard<std::mutex> lk(g_mu);
  g_runtime.push_back(Entry{op, M, N, K, epilogue, accumulate ? 1 : 0, tile, splits});
  g_memo.clear();
  return BLM_OK;
}

extern "C" int blm_gemm_plan_clear(int keep_builtin) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_runtime.clear();
  g_comm_runtime.clear();
  g_builtin_on = g_comm_builtin_on = keep_builtin != 0;
  g_memo.clear();
  return BLM_OK;
}

extern "C" int blm_gemm_plan_set_cus(int cus) {
  if (cus != 0 && (cus < 8 || cus > kChipCUs)) return blm_fail(BLM_ERR_INVALID, "blm_gemm_plan_set_cus: 0 (the whole chip) or 8..%d", kChipCUs);
  std::lock_guard<std::mutex> lk(g_mu);
  const int n = cus == 0 ? kChipCUs : cus;
  g_cus = n;
  return BLM_OK;
}

extern "C" int blm_gemm_plan_get_cus(void) { return plan_cus(); }

extern "C" int blm_gemm_plan_comm_window(float us) {
  if (!(us >= 0.f) || us > 1e7f) return blm_fail(BLM_ERR_INVALID, "blm_gemm_plan_comm_window: 0 (close) or a time in microseconds");
  std::lock_guard<std::mutex> lk(g_mu);
  if (us == 0.f) g_window_us = 0.0;
  else g_window_us = (g_window_us > 0.0 ? g_window_us : 0.0) + us;
  return BLM_OK;
}

extern "C" float blm_gemm_plan_comm_window_left(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  return g_window_us > 0.0 ? (float)g_window_us : 0.f;
}

extern "C" int blm_gemm_plan_set_comm(int op, int M, int N, int K, int epilogue, int accumulate, int tile, int splits) {
  if (op < BLM_GEMM_NT || op > BLM_GEMM_TN || !valid_tile(tile) || splits < -64 || splits > 64 || splits == 0 || splits == -1)
    return blm_fail(BLM_ERR_INVALID, "blm_gemm_plan_set_comm: bad op, tile or split count");
  std::lock_guard<std::mutex> lk(g_mu);
  g_comm_runtime.push_back(Entry{op, M, N, K, epilogue, accumulate ? 1 : 0, tile, splits});
  g_memo.clear();
  return BLM_OK;
}
