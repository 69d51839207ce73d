// dnaf_api.hip -- C ABI (include/dnaf.h) over the gfx950 row-generation kernels.
//
// Host responsibilities, all one-shot per table: integer thresholds from the
// float64 cumulative frequencies, forced-set slots, text layout (row offsets
// by prefix sum), sample nibble tables.  dnaf_generate() itself only
// launches the kernel on the context's stream.
#include "../../include/dnaf.h"
#include "rowgen_kernels.hip.h"
#include "rowrun_kernels.hip.h"
#include "rowmix_kernels.hip.h"
#include "bgzf_kernels.hip.h"
#include "gtz_kernels.hip.h"
#include "bed_kernels.hip.h"
#include "mt_jump.h"
#include "mt_kernels.hip.h"

#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

using dnaf::GenParams;
using dnaf::RowRec;

// Workgroups of rowrun_kernel per CU, capped through unused dynamic LDS: 12,288 static + 41,000 pad = 3 workgroups
// (12 waves) per CU instead of the 6 its registers allow.  The kernel is bound by its vector-instruction count and three
// waves per SIMD cover the instruction latencies; with fewer waves in flight each wave's consecutive text stores are closer
// in time, so fewer 128-byte lines that two of them share leave the L2 half-written -- measured on boxes in their slow
// state: 4.35 ms per launch at 6 workgroups per CU, 4.16 at 4, 4.13 at 3, 4.35 at 2 (profiles/r02_occupancy_sweep.txt).
#ifndef DNAF_RUN_LDS_PAD_DEFAULT
#define DNAF_RUN_LDS_PAD_DEFAULT 41000
#endif

struct dnaf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> evs; // event pairs, one per kernel launch since dnaf_timing_begin
    size_t n_timed = 0;          // launches recorded
    uint64_t n_launches = 0;     // launches since dnaf_timing_begin (recorded or not)
    std::string err;

    // samples
    bool have_samples = false;
    uint32_t S = 0, G = 0, W = 0, Wp = 0, n_male = 0; // Wp = packed row pitch on the device, in words (>= W)
    uint64_t x_row_bytes = 0; // cells of an X row
    uint8_t* d_male_nib = nullptr;
    uint32_t* d_ctrl2 = nullptr; // [W] 2-bit-domain mask of the control samples
    uint32_t* d_counts = nullptr;
    uint64_t counts_cap = 0;
    uint32_t* d_xoff = nullptr;
    uint32_t* d_xgroup = nullptr;      // [G] X rows, per group: cell offset | bytes << 16 | male nibble << 24 (rowmix_kernel)
    std::vector<uint8_t> h_flags;

    // forced set
    uint64_t n_slots = 0;
    uint8_t* d_forced = nullptr;
    std::unordered_map<int64_t, int32_t> slot_of;

    // rng
    uint32_t rng_mode = DNAF_RNG_PHILOX;
    uint64_t seed = 0, rng_row0 = 0;
    const char* last_kernel = ""; // name of the row kernel the last dnaf_generate launched (dnaf_last_kernel)
    uint32_t rng_div = 1; // table rows k * rng_div .. + rng_div - 1 share RNG row k (dnaf_set_rng_divisor)

    // snp table
    bool have_snps = false;
    uint64_t n_rows = 0;
    RowRec* d_recs = nullptr;
    uint8_t* d_prefix = nullptr;
    uint8_t* d_prefix_fixed = nullptr; // short rows: the prefixes once more, prefix_stride bytes apart (rowmix_kernel)
    uint32_t prefix_stride = 0;
    std::vector<uint64_t> h_text_off; // [n_rows+1]

    // outputs
    uint8_t* d_text = nullptr;
    uint64_t text_cap = 0;
    uint8_t* d_packed = nullptr;
    uint64_t packed_cap = 0;
    uint64_t last_begin = 0, last_rows = 0;
    uint32_t last_what = 0;
    bool have_output = false;

    // numpy-compatible MT19937 mode
    uint32_t mt_key[624] = {0};          // the worker's entry state: numpy.random.get_state()[1]
    uint32_t mt_pos = 624;               // ... and [2]
    std::vector<uint64_t> h_thr53;       // [n_rows][3] thresholds on the 53-bit roll numerator
    bool thr53_ok = true;                // every row's last cumulative frequency covers every 53-bit roll
    uint64_t* d_thr53 = nullptr;
    uint32_t* d_mt_base = nullptr;       // entry window on the device
    uint32_t* d_mt_states = nullptr;     // [mt_cap][624] block-aligned windows of waves [mt_lo, mt_lo + mt_n)
    uint32_t* d_mt_tmp = nullptr;
    uint64_t mt_cap = 0, mt_lo = 0, mt_n = 0;
    bool mt_valid = false;
    std::map<uint64_t, uint32_t*> mt_polys; // jump distance in words -> device list of the polynomial's set bits (mt_jump_kernel)

    // BGZF encoder
    dnaf::BgzfTables* d_tables = nullptr;
    bool have_tables = false;
    uint32_t* d_hist = nullptr;
    uint32_t* d_sizes = nullptr;
    uint16_t* d_thread_bits = nullptr;
    uint32_t* d_crcs = nullptr;
    uint32_t* d_large = nullptr;
    uint32_t* d_dense = nullptr;    // dnaf_read_packed: the matrix without its row pitch
    uint64_t dense_cap = 0;
    uint64_t* d_scan_u64 = nullptr; // per-workgroup totals of the member scans
    uint32_t* d_scan_u32 = nullptr;
    uint64_t* d_offsets = nullptr;
    uint64_t blocks_cap = 0;
    uint8_t* d_bgzf = nullptr;
    uint64_t bgzf_cap = 0, bgzf_bytes = 0;
    bool have_bgzf = false;
    hipEvent_t zev[2] = {nullptr, nullptr};
    float bgzf_ms = 0.f;
    // ... its genotype-domain part (members made from the packed matrix)
    dnaf::GtzTables* d_gtz_tables = nullptr;
    uint32_t* d_gtz_hist = nullptr;
    uint32_t* d_member_row = nullptr;
    uint8_t* d_fast = nullptr;
    uint32_t* d_todo = nullptr;
    uint8_t* d_stage = nullptr;
    uint64_t stage_cap = 0;
    bool have_gtz_tables = false;

    // PLINK .bed rows of the last packed matrix
    uint8_t* d_bed = nullptr;
    uint64_t bed_cap = 0, bed_bytes = 0;
    bool have_bed = false;

    // pinned host staging for device -> host copies at PCIe rate
    static constexpr int N_PINNED = 4;      // used in turn: a fetched range stays valid through the next three fetches
    void* h_pinned[N_PINNED] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t pinned_cap[N_PINNED] = {0, 0, 0, 0};
    int pinned_next = 0;

    // tuning
    uint32_t seg_iters = 32;
    bool force_segments = false;     // DNAF_FORCE_SEGMENTS=1: always the segment form of the kernel (A/B runs, tests)
    uint64_t min_run_waves = 8192;   // fewer runs than this and long rows: the segment form fills the GPU better
    uint32_t mix_x_max_groups = 512; // DNAF_MIX_X_MAX_GROUPS (tests / A-B): X, Y, MT rows in the short-row kernel's trips up to this many groups
    bool no_mix_x = false;           // DNAF_NO_MIX_X: X rows of short-row shapes a row at a time (tests / A-B)
    uint32_t force_rpw = 0;          // DNAF_ROWS_PER_WAVE: rows per wave of the line-owning form (tests)
    bool no_rowmix = false;          // DNAF_NO_ROWMIX=1: never the short-row form (A/B runs)
    uint32_t mix_max_groups = 509;   // rows of fewer groups than this take the short-row form: every shape with a dense packed pitch
    uint32_t run_lds_pad = DNAF_RUN_LDS_PAD_DEFAULT; // unused dynamic LDS per workgroup of rowrun_kernel: caps the workgroups per CU
};

static const size_t MAX_TIMED = 4096;
static thread_local std::string g_create_err = "";

static int fail(dnaf_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_create_err = buf;
    return code;
}

#define HIP_TRY(c, expr)                                                                                     \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail((c), e_ == hipErrorOutOfMemory ? DNAF_ERR_OOM : DNAF_ERR_HIP, "%s: %s", #expr,       \
                        hipGetErrorString(e_));                                                              \
    } while (0)

template <class T>
static void dev_free(T*& p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}

extern "C" {

int dnaf_abi_version(void) { return DNAF_ABI_VERSION; }

int dnaf_device_count(int* count)
{
    if (!count) return DNAF_ERR_BAD_ARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(nullptr, DNAF_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return DNAF_OK;
}

int dnaf_create(int device, dnaf_ctx** out)
{
    if (!out) return fail(nullptr, DNAF_ERR_BAD_ARG, "dnaf_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, DNAF_ERR_HIP, "dnaf_create: no HIP device (%s)", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(nullptr, DNAF_ERR_BAD_ARG, "dnaf_create: device %d of %d", device, n);
    dnaf_ctx* c = new (std::nothrow) dnaf_ctx();
    if (!c) return fail(nullptr, DNAF_ERR_OOM, "dnaf_create: host allocation");
    c->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreate(&c->stream)) != hipSuccess) {
        delete c;
        return fail(nullptr, DNAF_ERR_HIP, "dnaf_create: %s", hipGetErrorString(e));
    }
    if (const char* s = getenv("DNAF_FORCE_SEGMENTS")) c->force_segments = atoi(s) != 0;
    if (const char* s = getenv("DNAF_MIN_RUN_WAVES")) c->min_run_waves = strtoull(s, nullptr, 10);
    if (const char* s = getenv("DNAF_NO_ROWMIX")) c->no_rowmix = atoi(s) != 0;
    if (const char* s = getenv("DNAF_MIX_MAX_GROUPS")) c->mix_max_groups = std::min<uint32_t>((uint32_t)strtoul(s, nullptr, 10), dnaf::MIX_MAX_GROUPS);
    if (const char* s = getenv("DNAF_RUN_LDS_PAD")) c->run_lds_pad = (uint32_t)std::min<unsigned long long>(strtoull(s, nullptr, 10), 48u << 10);
    if (const char* s = getenv("DNAF_NO_MIX_X")) c->no_mix_x = atoi(s) != 0;
    if (const char* s = getenv("DNAF_MIX_X_MAX_GROUPS")) c->mix_x_max_groups = (uint32_t)strtoul(s, nullptr, 10);
    if (const char* s = getenv("DNAF_ROWS_PER_WAVE")) c->force_rpw = (uint32_t)std::min<unsigned long long>(strtoull(s, nullptr, 10), 1u << 20);
    if (const char* s = getenv("DNAF_SEG_ITERS")) {
        int v = atoi(s);
        if (v >= 1 && v <= 4096) c->seg_iters = (uint32_t)v;
    }
    *out = c;
    return DNAF_OK;
}

int dnaf_destroy(dnaf_ctx* c)
{
    if (!c) return DNAF_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    dev_free(c->d_male_nib);
    dev_free(c->d_ctrl2);
    dev_free(c->d_xoff);
    dev_free(c->d_xgroup);
    dev_free(c->d_forced);
    dev_free(c->d_recs);
    dev_free(c->d_prefix);
    dev_free(c->d_prefix_fixed);
    dev_free(c->d_text);
    dev_free(c->d_packed);
    for (void* hp : c->h_pinned) if (hp) (void)hipHostFree(hp);
    dev_free(c->d_counts);
    dev_free(c->d_thr53);
    dev_free(c->d_mt_base);
    dev_free(c->d_mt_states);
    dev_free(c->d_mt_tmp);
    for (auto& kv : c->mt_polys) (void)hipFree(kv.second);
    dev_free(c->d_tables);
    dev_free(c->d_hist);
    dev_free(c->d_sizes);
    dev_free(c->d_thread_bits);
    dev_free(c->d_crcs);
    dev_free(c->d_large);
    dev_free(c->d_scan_u64);
    dev_free(c->d_scan_u32);
    dev_free(c->d_dense);
    dev_free(c->d_offsets);
    dev_free(c->d_bgzf);
    dev_free(c->d_bed);
    dev_free(c->d_gtz_tables);
    dev_free(c->d_gtz_hist);
    dev_free(c->d_member_row);
    dev_free(c->d_fast);
    dev_free(c->d_todo);
    dev_free(c->d_stage);
    for (hipEvent_t e : c->zev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->evs) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return DNAF_OK;
}

const char* dnaf_last_error(const dnaf_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int dnaf_set_samples(dnaf_ctx* c, const uint8_t* flags, uint64_t n_samples)
{
    if (!c) return DNAF_ERR_BAD_ARG;
    if (n_samples && !flags) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_set_samples: flags is NULL");
    if (n_samples > 0x3FFFFFFFull) return fail(c, DNAF_ERR_UNSUPPORTED, "dnaf_set_samples: more than 2^30-1 samples");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->have_samples = c->have_snps = c->have_output = false;
    c->mt_valid = false; // wave start states depend on the number of samples
    c->n_slots = 0;
    c->slot_of.clear();
    dev_free(c->d_forced);
    dev_free(c->d_male_nib);
    dev_free(c->d_ctrl2);
    dev_free(c->d_xoff);
    dev_free(c->d_xgroup);
    c->S = (uint32_t)n_samples;
    c->G = (c->S + 3u) / 4u;
    c->W = (c->S + 15u) / 16u;
    // device pitch of the packed matrix: whole 128-byte lines per row once a row is long enough for that to cost little
    // (every line of the matrix then has one writer, rowrun_kernels.hip.h); dense below
    c->Wp = c->W >= 128u ? (c->W + 31u) / 32u * 32u : c->W;
    c->h_flags.assign(flags, flags + n_samples);
    std::vector<uint8_t> nib(c->G ? c->G : 1, 0);
    std::vector<uint32_t> xoff((size_t)c->G + 1, 0);
    uint32_t n_male = 0;
    uint64_t off = 0;
    for (uint32_t i = 0; i < c->S; ++i) {
        if ((i & 3u) == 0) xoff[i >> 2] = (uint32_t)off;
        const bool male = flags[i] & DNAF_SAMPLE_MALE;
        if (male) { nib[i >> 2] |= (uint8_t)(1u << (i & 3u)); ++n_male; }
        off += male ? 2 : 4;
        if (off > 0xFFFFFFF0ull) return fail(c, DNAF_ERR_UNSUPPORTED, "dnaf_set_samples: row longer than 4 GiB");
    }
    xoff[c->G] = (uint32_t)off;
    c->n_male = n_male;
    c->x_row_bytes = off;
    {
        std::vector<uint32_t> ctrl2(c->W ? c->W : 1, 0);
        for (uint32_t i = 0; i < c->S; ++i)
            if (flags[i] & DNAF_SAMPLE_CONTROL) ctrl2[i >> 4] |= 3u << (2u * (i & 15u));
        HIP_TRY(c, hipMalloc((void**)&c->d_ctrl2, ctrl2.size() * sizeof(uint32_t)));
        HIP_TRY(c, hipMemcpy(c->d_ctrl2, ctrl2.data(), ctrl2.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    HIP_TRY(c, hipMalloc((void**)&c->d_male_nib, nib.size()));
    HIP_TRY(c, hipMalloc((void**)&c->d_xoff, xoff.size() * sizeof(uint32_t)));
    HIP_TRY(c, hipMemcpy(c->d_male_nib, nib.data(), nib.size(), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_xoff, xoff.data(), xoff.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    {
        // what the short-row kernel needs of an X row's group in one word (used while rows are short: offsets below 64 KiB)
        std::vector<uint32_t> xg(c->G ? c->G : 1, 0);
        for (uint32_t g = 0; g < c->G; ++g) xg[g] = (xoff[g] & 0xFFFFu) | (((xoff[g + 1] - xoff[g]) & 31u) << 16) | ((uint32_t)nib[g] << 24);
        HIP_TRY(c, hipMalloc((void**)&c->d_xgroup, xg.size() * sizeof(uint32_t)));
        HIP_TRY(c, hipMemcpy(c->d_xgroup, xg.data(), xg.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    c->have_samples = true;
    return DNAF_OK;
}

int dnaf_set_forced(dnaf_ctx* c, const int64_t* snp_ids, uint64_t n_slots, const uint64_t* carrier_off,
                    const uint32_t* carrier_idx)
{
    if (!c) return DNAF_ERR_BAD_ARG;
    if (!c->have_samples) return fail(c, DNAF_ERR_STATE, "dnaf_set_forced: call dnaf_set_samples first");
    if (n_slots && (!snp_ids || !carrier_off)) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_set_forced: NULL argument");
    if (n_slots > 0x7FFFFFFFull) return fail(c, DNAF_ERR_UNSUPPORTED, "dnaf_set_forced: too many slots");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->have_snps = c->have_output = false; // rows' slots are resolved in dnaf_set_snps
    c->n_slots = 0;
    c->slot_of.clear();
    dev_free(c->d_forced);
    if (n_slots == 0) return DNAF_OK;
    const uint64_t G = c->G ? c->G : 1;
    std::vector<uint8_t> nib;
    try {
        nib.assign((size_t)(n_slots * G), 0);
    } catch (const std::bad_alloc&) {
        return fail(c, DNAF_ERR_OOM, "dnaf_set_forced: host allocation of %llu bytes", (unsigned long long)(n_slots * G));
    }
    for (uint64_t k = 0; k < n_slots; ++k) {
        if (carrier_off[k + 1] < carrier_off[k]) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_set_forced: carrier_off not monotone");
        if (c->slot_of.count(snp_ids[k])) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_set_forced: duplicate snp id %lld", (long long)snp_ids[k]);
        c->slot_of[snp_ids[k]] = (int32_t)k;
        for (uint64_t j = carrier_off[k]; j < carrier_off[k + 1]; ++j) {
            const uint32_t i = carrier_idx[j];
            if (i >= c->S) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_set_forced: sample index %u out of range", i);
            if (c->h_flags[i] & DNAF_SAMPLE_CONTROL)
                return fail(c, DNAF_ERR_BAD_ARG, "dnaf_set_forced: sample %u is a control (never forced, pop_factory.py:485)", i);
            nib[(size_t)(k * G + (i >> 2))] |= (uint8_t)(1u << (i & 3u));
        }
    }
    HIP_TRY(c, hipMalloc((void**)&c->d_forced, nib.size()));
    HIP_TRY(c, hipMemcpy(c->d_forced, nib.data(), nib.size(), hipMemcpyHostToDevice));
    c->n_slots = n_slots;
    return DNAF_OK;
}

int dnaf_set_rng(dnaf_ctx* c, uint32_t mode, uint64_t seed, uint64_t rng_row0)
{
    if (!c) return DNAF_ERR_BAD_ARG;
    if (mode != DNAF_RNG_PHILOX && mode != DNAF_RNG_MT19937) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_set_rng: unknown mode %u", mode);
    if (mode == DNAF_RNG_MT19937) {
        // numpy.random.seed(int): init_genrand, valid for 0 .. 2^32-1 (pop_factory.py:235)
        if (seed > 0xFFFFFFFFull) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_set_rng: an MT19937 seed must be below 2^32 (numpy.random.seed)");
        if (c->rng_mode != mode || c->seed != seed || c->mt_pos != 624) {
            dnaf::mt_init_genrand((uint32_t)seed, c->mt_key);
            c->mt_pos = 624;
            c->mt_valid = false;
            dev_free(c->d_mt_base);
        }
    }
    c->rng_mode = mode;
    c->seed = seed;
    c->rng_row0 = rng_row0;
    return DNAF_OK;
}

int dnaf_set_mt_state(dnaf_ctx* c, const uint32_t* key, uint32_t pos, uint64_t rng_row0)
{
    if (!c) return DNAF_ERR_BAD_ARG;
    if (!key || pos > 624) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_set_mt_state: key is NULL or pos > 624");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    memcpy(c->mt_key, key, sizeof c->mt_key);
    c->mt_pos = pos;
    c->mt_valid = false;
    dev_free(c->d_mt_base);
    c->rng_mode = DNAF_RNG_MT19937;
    c->seed = 0xFFFFFFFFFFFFFFFFull; // an explicit state, not a seed
    c->rng_row0 = rng_row0;
    return DNAF_OK;
}

int dnaf_set_rng_row0(dnaf_ctx* c, uint64_t rng_row0)
{
    if (!c) return DNAF_ERR_BAD_ARG;
    c->rng_row0 = rng_row0;
    return DNAF_OK;
}

int dnaf_set_rng_divisor(dnaf_ctx* c, uint32_t n)
{
    if (!c) return DNAF_ERR_BAD_ARG;
    if (n < 1) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_set_rng_divisor: n must be at least 1");
    c->rng_div = n;
    return DNAF_OK;
}

// floor(cum * 2^53) clamped to 2^53: `cum >= r * 2^-53` <=> r <= T for the numerators r < 2^53 of numpy's doubles
static uint64_t threshold_u53(double cum)
{
    const double s = std::floor(cum * 9007199254740992.0);
    return s >= 9007199254740992.0 ? (1ull << 53) : (uint64_t)s;
}

// floor(cum * 2^32) clamped to u32: `cum >= u * 2^-32` <=> u <= T (both sides exact in float64)
static bool threshold_u32(double cum, uint32_t* t)
{
    if (!(cum >= 0.0)) return false; // negative or NaN
    const double s = std::floor(cum * 4294967296.0);
    *t = s >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)s;
    return true;
}

int dnaf_set_snps(dnaf_ctx* c, const dnaf_snp_table* t)
{
    if (!c) return DNAF_ERR_BAD_ARG;
    if (!c->have_samples) return fail(c, DNAF_ERR_STATE, "dnaf_set_snps: call dnaf_set_samples first");
    if (!t) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_set_snps: table is NULL");
    const uint64_t n = t->n_rows;
    if (n && (!t->snp_id || !t->chrom_class || !t->n_alleles || !t->cum_freq || !t->prefix_off))
        return fail(c, DNAF_ERR_BAD_ARG, "dnaf_set_snps: NULL column");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->have_snps = c->have_output = false;
    dev_free(c->d_recs);
    dev_free(c->d_prefix);
    dev_free(c->d_prefix_fixed);
    dev_free(c->d_thr53);
    c->thr53_ok = true;
    std::vector<RowRec> recs;
    try {
        recs.resize((size_t)n);
        c->h_text_off.assign((size_t)n + 1, 0);
        c->h_thr53.assign((size_t)n * 3, 1ull << 53);
    } catch (const std::bad_alloc&) {
        return fail(c, DNAF_ERR_OOM, "dnaf_set_snps: host allocation for %llu rows", (unsigned long long)n);
    }
    const uint64_t S = c->S;
    const uint64_t cells_len[4] = {S ? 4 * S : 1, S ? c->x_row_bytes : 1, S ? 2 * S : 1, S ? 2 * S : 1};
    const uint64_t pbase = n ? t->prefix_off[0] : 0;
    uint64_t off = 0, max_plen = 0;
    c->prefix_stride = 0;
    for (uint64_t j = 0; j < n; ++j) {
        RowRec& r = recs[(size_t)j];
        const uint32_t na = t->n_alleles[j], cls = t->chrom_class[j];
        if (na < 1 || na > DNAF_MAX_ALLELES)
            return fail(c, DNAF_ERR_UNSUPPORTED, "dnaf_set_snps: row %llu has %u alleles (1..%d supported)", (unsigned long long)j, na, DNAF_MAX_ALLELES);
        if (cls > DNAF_CLASS_MT) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_set_snps: row %llu chrom_class %u", (unsigned long long)j, cls);
        uint32_t thr[DNAF_MAX_ALLELES] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        for (uint32_t k = 0; k < na; ++k)
            if (!threshold_u32(t->cum_freq[j * DNAF_MAX_ALLELES + k], &thr[k]))
                return fail(c, DNAF_ERR_BAD_ARG, "dnaf_set_snps: row %llu cumulative frequency %u is negative or NaN", (unsigned long long)j, k);
        if (S && thr[na - 1] != 0xFFFFFFFFu)
            return fail(c, DNAF_ERR_THRESHOLD, "dnaf_set_snps: row %llu last cumulative frequency %.17g < 1: a roll above it has no allele "
                        "(the reference raises TypeError, pop_factory.py:489)", (unsigned long long)j, t->cum_freq[j * DNAF_MAX_ALLELES + na - 1]);
        r.thr[0] = thr[0]; r.thr[1] = thr[1]; r.thr[2] = thr[2];
        for (uint32_t k = 0; k < na && k < 3; ++k) c->h_thr53[(size_t)j * 3 + k] = threshold_u53(t->cum_freq[j * DNAF_MAX_ALLELES + k]);
        if (S && threshold_u53(t->cum_freq[j * DNAF_MAX_ALLELES + na - 1]) < (1ull << 53) - 1) c->thr53_ok = false;
        if (t->prefix_off[j + 1] < t->prefix_off[j]) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_set_snps: prefix_off not monotone at row %llu", (unsigned long long)j);
        const uint64_t plen = t->prefix_off[j + 1] - t->prefix_off[j], poff = t->prefix_off[j] - pbase;
        if (plen > 0xFFFFu || poff > 0xFFFFFFFFull)
            return fail(c, DNAF_ERR_UNSUPPORTED, "dnaf_set_snps: row %llu prefix too long / prefix blob over 4 GiB", (unsigned long long)j);
        r.prefix_off = (uint32_t)poff;
        r.prefix_len = (uint16_t)plen;
        max_plen = std::max(max_plen, plen);
        r.text_off = off;
        auto it = c->n_slots ? c->slot_of.find(t->snp_id[j]) : c->slot_of.end();
        r.slot = it == c->slot_of.end() ? -1 : it->second;
        r.cls = (uint8_t)cls;
        r.n_alleles = (uint8_t)na;
        c->h_text_off[(size_t)j] = off;
        off += plen + cells_len[cls];
    }
    c->h_text_off[(size_t)n] = off;
    const uint64_t pbytes = n ? t->prefix_off[n] - pbase : 0;
    if (pbytes && !t->prefix_bytes) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_set_snps: prefix_bytes is NULL");
    {
        hipError_t e = hipMalloc((void**)&c->d_recs, (n ? n : 1) * sizeof(RowRec));
        if (e == hipSuccess) e = hipMalloc((void**)&c->d_prefix, pbytes ? pbytes : 1);
        if (e == hipSuccess && n) e = hipMemcpy(c->d_recs, recs.data(), n * sizeof(RowRec), hipMemcpyHostToDevice);
        if (e == hipSuccess && pbytes) e = hipMemcpy(c->d_prefix, t->prefix_bytes + pbase, pbytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) { // no half-loaded table is left behind
            dev_free(c->d_recs);
            dev_free(c->d_prefix);
    dev_free(c->d_prefix_fixed);
            return fail(c, e == hipErrorOutOfMemory ? DNAF_ERR_OOM : DNAF_ERR_HIP, "dnaf_set_snps: %s", hipGetErrorString(e));
        }
    }
    // short rows: a second copy of the fixed columns at a fixed stride, so that a wave of rowmix_kernel can ask for them
    // without having its rows' records first (rowmix_kernels.hip.h).  Skipped when a prefix is longer than 64 bytes.
    if (n && c->S && c->G < c->mix_max_groups && c->Wp == c->W && max_plen <= 64) {
        const uint32_t stride = (uint32_t)std::max<uint64_t>(16, (max_plen + 15) / 16 * 16);
        if (hipMalloc((void**)&c->d_prefix_fixed, n * stride + 64) == hipSuccess) {
            hipLaunchKernelGGL(dnaf::prefix_fix_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, (const RowRec*)c->d_recs,
                               (const uint8_t*)c->d_prefix, c->d_prefix_fixed, n, stride);
            if (hipGetLastError() == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess) c->prefix_stride = stride;
        }
        if (!c->prefix_stride) { // optional: the kernel works without it
            (void)hipGetLastError();
            dev_free(c->d_prefix_fixed);
        }
    }
    c->n_rows = n;
    c->have_snps = true;
    return DNAF_OK;
}

static int check_range(const dnaf_ctx* c, uint64_t b, uint64_t n)
{
    if (!c || !c->have_snps) return DNAF_ERR_STATE;
    if (b > c->n_rows || n > c->n_rows - b) return DNAF_ERR_BAD_ARG;
    return DNAF_OK;
}

int dnaf_text_bytes(const dnaf_ctx* c, uint64_t row_begin, uint64_t n_rows, uint64_t* bytes)
{
    if (!c || !bytes) return DNAF_ERR_BAD_ARG;
    if (int rc = check_range(c, row_begin, n_rows))
        return fail(const_cast<dnaf_ctx*>(c), rc, "dnaf_text_bytes: no SNP table or rows [%llu,+%llu) out of range", (unsigned long long)row_begin, (unsigned long long)n_rows);
    *bytes = c->h_text_off[(size_t)(row_begin + n_rows)] - c->h_text_off[(size_t)row_begin];
    return DNAF_OK;
}

int dnaf_row_offsets(const dnaf_ctx* c, uint64_t row_begin, uint64_t n_rows, uint64_t* off)
{
    if (!c || !off) return DNAF_ERR_BAD_ARG;
    if (int rc = check_range(c, row_begin, n_rows))
        return fail(const_cast<dnaf_ctx*>(c), rc, "dnaf_row_offsets: no SNP table or rows [%llu,+%llu) out of range", (unsigned long long)row_begin, (unsigned long long)n_rows);
    const uint64_t base = c->h_text_off[(size_t)row_begin];
    for (uint64_t j = 0; j <= n_rows; ++j) off[j] = c->h_text_off[(size_t)(row_begin + j)] - base;
    return DNAF_OK;
}

static int grow_outputs(dnaf_ctx* c, uint64_t tbytes, uint64_t pbytes)
{
    if (tbytes > c->text_cap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        dev_free(c->d_text);
        c->text_cap = 0;
        c->have_output = false;
        HIP_TRY(c, hipMalloc((void**)&c->d_text, tbytes + 64)); // slack: nothing is stored past tbytes
        c->text_cap = tbytes;
    }
    if (pbytes > c->packed_cap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        dev_free(c->d_packed);
        c->packed_cap = 0;
        c->have_output = false;
        HIP_TRY(c, hipMalloc((void**)&c->d_packed, pbytes + 64));
        c->packed_cap = pbytes;
    }
    return DNAF_OK;
}

int dnaf_reserve_output(dnaf_ctx* c, uint64_t text_bytes, uint64_t packed_words)
{
    if (!c) return DNAF_ERR_BAD_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    // packed_words counts dense rows of W words; the device rows are Wp words apart
    const uint64_t rows = c->W ? (packed_words + c->W - 1) / c->W : 0;
    return grow_outputs(c, text_bytes, rows * (uint64_t)c->Wp * 4u);
}

// ---- numpy-compatible MT19937 mode (mt_kernels.hip.h) ----
// One jumper per process, built once: contexts of several host threads (multi-GPU runs) reach this concurrently,
// and C++ guarantees that the initialisation of a function-local static runs exactly once and is complete before
// any caller proceeds.
static dnaf::MtJumper& mt_jumper()
{
    static dnaf::MtJumper j = [] { dnaf::MtJumper t; t.init(); return t; }();
    return j;
}

static uint64_t gcd_u64(uint64_t a, uint64_t b) { while (b) { const uint64_t t = a % b; a = b; b = t; } return a; }

// x^(J-1) mod phi as the list of set bits mt_jump_kernel reads (layout there)
static std::vector<uint32_t> mt_poly_list(const dnaf::MtPoly& g)
{
    std::vector<uint32_t> l(dnaf::MT_POLY_LIST_HEAD, 0u);
    l.reserve(dnaf::MT_POLY_LIST_MAX);
    for (uint32_t w = 0; w < 312u; ++w)
        for (uint64_t bits = g.w[w]; bits; bits &= bits - 1) l.push_back(4u * (64u * w + (uint32_t)__builtin_ctzll(bits)));
    while ((l.size() - dnaf::MT_POLY_LIST_HEAD) % 16u) l.push_back(4u * dnaf::MT_PAD_WORD);
    l[0] = (uint32_t)((l.size() - dnaf::MT_POLY_LIST_HEAD) / 16u);
    return l;
}

// device list of x^(J-1) mod phi for a jump of J >= 1 words (cached per distance)
static int mt_poly_for(dnaf_ctx* c, uint64_t J, const uint32_t** out)
{
    auto it = c->mt_polys.find(J);
    if (it == c->mt_polys.end()) {
        dnaf::MtPoly g;
        mt_jumper().xpow_mod(J - 1, &g);
        const std::vector<uint32_t> l = mt_poly_list(g);
        uint32_t* d = nullptr;
        HIP_TRY(c, hipMalloc((void**)&d, l.size() * sizeof(uint32_t)));
        hipError_t e = hipMemcpy(d, l.data(), l.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(d); return fail(c, DNAF_ERR_HIP, "mt_poly_for: %s", hipGetErrorString(e)); }
        it = c->mt_polys.emplace(J, d).first;
    }
    *out = it->second;
    return DNAF_OK;
}

// Make d_mt_states hold the block-aligned windows of waves [lo, lo + n): wave k starts at output word
// k * words_per_wave after the entry state, i.e. in block k * D + (pos == 624) of the word sequence.
static int mt_prepare_states(dnaf_ctx* c, uint64_t lo, uint64_t n, uint64_t D)
{
    if (c->mt_valid && c->mt_lo == lo && c->mt_n >= n) return DNAF_OK;
    if (n > c->mt_cap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        uint32_t *a = nullptr, *b = nullptr;
        HIP_TRY(c, hipMalloc((void**)&a, n * 624 * sizeof(uint32_t)));
        hipError_t e2 = hipMalloc((void**)&b, n * 624 * sizeof(uint32_t));
        if (e2 == hipSuccess && c->mt_valid && c->mt_n) e2 = hipMemcpy(a, c->d_mt_states, c->mt_n * 624 * sizeof(uint32_t), hipMemcpyDeviceToDevice);
        if (e2 != hipSuccess) { // nothing allocated here outlives the failure
            (void)hipFree(a);
            if (b) (void)hipFree(b);
            return fail(c, e2 == hipErrorOutOfMemory ? DNAF_ERR_OOM : DNAF_ERR_HIP, "mt_prepare_states: %s", hipGetErrorString(e2));
        }
        dev_free(c->d_mt_states);
        dev_free(c->d_mt_tmp);
        c->d_mt_states = a;
        c->d_mt_tmp = b;
        c->mt_cap = n;
    }
    if (!c->d_mt_base) {
        HIP_TRY(c, hipMalloc((void**)&c->d_mt_base, sizeof c->mt_key));
        HIP_TRY(c, hipMemcpy(c->d_mt_base, c->mt_key, sizeof c->mt_key, hipMemcpyHostToDevice));
    }
    const uint32_t* poly = nullptr;
    if (c->mt_valid && lo > c->mt_lo && n <= c->mt_n) {
        // a later window of the same stream -- the next tile of a sequential sweep, or of a device that takes every G-th
        // tile of a job: every cached state advances by the same distance (one polynomial per distance, cached)
        if (lo - c->mt_lo > 0xFFFFFFFFFFFFFFFFull / (D * 624)) return fail(c, DNAF_ERR_UNSUPPORTED, "dnaf_generate: MT19937 stream position overflows 64 bits");
        if (int rc = mt_poly_for(c, (lo - c->mt_lo) * D * 624, &poly)) return rc;
        hipLaunchKernelGGL(dnaf::mt_jump_kernel, dim3((uint32_t)c->mt_n), dim3(640), 0, c->stream, (const uint32_t*)c->d_mt_states,
                           (const uint32_t*)nullptr, c->d_mt_tmp, poly);
        HIP_TRY(c, hipGetLastError());
        std::swap(c->d_mt_states, c->d_mt_tmp);
        c->mt_lo = lo;
        return DNAF_OK;
    }
    // rebuild: first state by one jump from the entry window, the rest by doubling
    c->mt_valid = false;
    const uint64_t blk0 = lo * D + (c->mt_pos == 624 ? 1 : 0);
    if (blk0 == 0) {
        HIP_TRY(c, hipMemcpyAsync(c->d_mt_states, c->d_mt_base, sizeof c->mt_key, hipMemcpyDeviceToDevice, c->stream));
    } else {
        if (blk0 > 0xFFFFFFFFFFFFFFFFull / 624) return fail(c, DNAF_ERR_UNSUPPORTED, "dnaf_generate: MT19937 stream position overflows 64 bits");
        if (int rc = mt_poly_for(c, blk0 * 624, &poly)) return rc;
        hipLaunchKernelGGL(dnaf::mt_jump_kernel, dim3(1), dim3(640), 0, c->stream, (const uint32_t*)c->d_mt_base, (const uint32_t*)nullptr,
                           c->d_mt_states, poly);
        HIP_TRY(c, hipGetLastError());
    }
    for (uint64_t have = 1; have < n; have *= 2) {
        const uint64_t cnt = have < n - have ? have : n - have;
        if (int rc = mt_poly_for(c, have * D * 624, &poly)) return rc;
        hipLaunchKernelGGL(dnaf::mt_jump_kernel, dim3((uint32_t)cnt), dim3(640), 0, c->stream, (const uint32_t*)c->d_mt_states,
                           (const uint32_t*)nullptr, c->d_mt_states + have * 624, poly);
        HIP_TRY(c, hipGetLastError());
    }
    c->mt_lo = lo;
    c->mt_n = n;
    c->mt_valid = true;
    return DNAF_OK;
}

static int generate_mt(dnaf_ctx* c, uint64_t row_begin, uint64_t n_rows, uint32_t what)
{
    if (!c->thr53_ok)
        return fail(c, DNAF_ERR_THRESHOLD, "dnaf_generate: a row's last cumulative frequency is below 1: a 53-bit roll above it has no allele "
                    "(the reference raises TypeError, pop_factory.py:489)");
    const uint64_t S = c->S;
    const uint64_t R = 156 / gcd_u64(S, 156);   // rows per wave: 4*S*R is a multiple of 624
    const uint64_t D = 4 * S * R / 624;         // blocks per wave
    const uint64_t N = c->rng_div;              // table rows that share a stream row (1 ordinarily)
    if (row_begin % N)
        return fail(c, DNAF_ERR_UNSUPPORTED, "dnaf_generate: with an RNG divisor of %llu a range must start on a multiple of it (row %llu does not)",
                    (unsigned long long)N, (unsigned long long)row_begin);
    const uint64_t rng_row = c->rng_row0 + row_begin / N;
    const uint64_t stream_rows = (n_rows + N - 1) / N;
    if (rng_row % R)
        return fail(c, DNAF_ERR_UNSUPPORTED, "dnaf_generate: in MT19937 mode a range must start on a multiple of %llu rows "
                    "(RNG row %llu does not), so that every wave starts on a block of the stream", (unsigned long long)R, (unsigned long long)rng_row);
    const uint64_t n_waves = (stream_rows + R - 1) / R;
    if (n_waves > 0x7FFFFFFFull) return fail(c, DNAF_ERR_UNSUPPORTED, "dnaf_generate: too many waves; generate fewer rows per call");
    if (!c->d_thr53) {
        HIP_TRY(c, hipMalloc((void**)&c->d_thr53, c->h_thr53.size() * sizeof(uint64_t) + 8));
        HIP_TRY(c, hipMemcpy(c->d_thr53, c->h_thr53.data(), c->h_thr53.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    }
    const bool rec = c->n_timed < MAX_TIMED;
    if (rec && c->evs.size() < 2 * (c->n_timed + 1)) {
        hipEvent_t a = nullptr, b = nullptr;
        HIP_TRY(c, hipEventCreate(&a));
        c->evs.push_back(a);
        HIP_TRY(c, hipEventCreate(&b));
        c->evs.push_back(b);
    }
    if (rec) HIP_TRY(c, hipEventRecord(c->evs[2 * c->n_timed], c->stream));
    if (int rc = mt_prepare_states(c, rng_row / R, n_waves, D)) return rc;
    dnaf::MtGenParams mp{};
    GenParams& p = mp.g;
    p.recs = c->d_recs;
    p.prefix = c->d_prefix;
    p.prefix_fixed = c->d_prefix_fixed;
    p.prefix_stride = c->prefix_stride;
    p.male_nib = c->d_male_nib;
    p.xoff = c->d_xoff;
    p.forced = c->d_forced;
    p.text = c->d_text;
    p.packed = c->d_packed;
    p.packed_pitch = (uint64_t)c->Wp * 4u;
    p.text_base = c->h_text_off[(size_t)row_begin];
    p.row_begin = row_begin;
    p.n_rows = n_rows;
    p.rng_row0 = c->rng_row0;
    p.rng_div = c->rng_div;
    p.key0 = p.key1 = 0;
    p.S = c->S; p.G = c->G; p.W = c->W;
    p.nseg = 1; p.seg_groups = 64; p.rows_per_wave = 1;
    mp.thr53 = c->d_thr53;
    mp.states = c->d_mt_states;
    mp.start_off = c->mt_pos == 624 ? 0 : c->mt_pos;
    mp.rows_per_wave = (uint32_t)R;
    mp.wave0_row = 0;
    mp.row_stride = (uint32_t)N;
    const dim3 block(64);
    for (uint64_t phase = 0; phase < N && phase < n_rows; ++phase) { // one launch per stripe of rows that draw the stream afresh
        mp.row_phase = (uint32_t)phase;
        mp.n_stream_rows = (n_rows - phase + N - 1) / N;
        const dim3 grid((uint32_t)((mp.n_stream_rows + R - 1) / R));
        if (what == (DNAF_OUT_TEXT | DNAF_OUT_PACKED)) hipLaunchKernelGGL((dnaf::mt_rowgen_kernel<true, true>), grid, block, 0, c->stream, mp);
        else if (what == DNAF_OUT_TEXT) hipLaunchKernelGGL((dnaf::mt_rowgen_kernel<true, false>), grid, block, 0, c->stream, mp);
        else hipLaunchKernelGGL((dnaf::mt_rowgen_kernel<false, true>), grid, block, 0, c->stream, mp);
    }
    HIP_TRY(c, hipGetLastError());
    c->last_kernel = "mt_rowgen_kernel";
    if (rec) {
        HIP_TRY(c, hipEventRecord(c->evs[2 * c->n_timed + 1], c->stream));
        ++c->n_timed;
    }
    ++c->n_launches;
    return DNAF_OK;
}

int dnaf_last_kernel(dnaf_ctx* c, char* name, uint64_t cap)
{
    if (!c || !name || !cap) return DNAF_ERR_BAD_ARG;
    snprintf(name, (size_t)cap, "%s", c->last_kernel);
    return DNAF_OK;
}

int dnaf_generate(dnaf_ctx* c, uint64_t row_begin, uint64_t n_rows, uint32_t what)
{
    if (!c) return DNAF_ERR_BAD_ARG;
    if (int rc = check_range(c, row_begin, n_rows))
        return fail(c, rc, "dnaf_generate: no SNP table or rows [%llu,+%llu) out of range", (unsigned long long)row_begin, (unsigned long long)n_rows);
    if (!(what & (DNAF_OUT_TEXT | DNAF_OUT_PACKED)) || (what & ~(DNAF_OUT_TEXT | DNAF_OUT_PACKED)))
        return fail(c, DNAF_ERR_BAD_ARG, "dnaf_generate: what = %u", what);
    HIP_TRY(c, hipSetDevice(c->device));
    const uint64_t tbytes = c->h_text_off[(size_t)(row_begin + n_rows)] - c->h_text_off[(size_t)row_begin];
    const uint64_t pbytes = n_rows * (uint64_t)c->Wp * 4u;
    if (int rc = grow_outputs(c, (what & DNAF_OUT_TEXT) ? tbytes : 0, (what & DNAF_OUT_PACKED) ? pbytes : 0)) return rc;
    c->last_begin = row_begin;
    c->last_rows = n_rows;
    c->last_what = what;
    c->have_output = true;
    c->have_bgzf = false;
    if (n_rows == 0) return DNAF_OK;
    if (c->rng_mode == DNAF_RNG_MT19937 && c->S) return generate_mt(c, row_begin, n_rows, what); // S == 0 draws nothing

    GenParams p{};
    p.recs = c->d_recs;
    p.prefix = c->d_prefix;
    p.prefix_fixed = c->d_prefix_fixed;
    p.prefix_stride = c->prefix_stride;
    p.male_nib = c->d_male_nib;
    p.xoff = c->d_xoff;
    p.forced = c->d_forced;
    p.text = c->d_text;
    p.packed = c->d_packed;
    p.packed_pitch = (uint64_t)c->Wp * 4u;
    p.text_base = c->h_text_off[(size_t)row_begin];
    p.row_begin = row_begin;
    p.n_rows = n_rows;
    p.rng_row0 = c->rng_row0;
    p.rng_div = c->rng_div;
    p.key0 = (uint32_t)c->seed;
    p.key1 = (uint32_t)(c->seed >> 32);
    p.S = c->S; p.G = c->G; p.W = c->W;
    if (c->G) { // rowmix_kernel's divisions, done here once instead of by every wave
        p.mix_gmagic = (uint32_t)std::min<uint64_t>(0x100000000ull / c->G, 0xFFFFFFFFull);
        p.mix_step_rows = 64u / c->G;
        p.mix_step_groups = 64u % c->G;
        // X / Y / MT rows in the lane-spanning trips: +12 % at 200 samples, +7 % at 400, +3 % at 1,000, within the noise at 1,600 and
        // 2,000 (profiles/r03_shapes.txt)
        p.mix_x_rows = (!c->no_mix_x && c->G <= c->mix_x_max_groups) ? 1u : 0u;
        p.xgroup = c->d_xgroup;
    }
    // Two forms of the kernel.  Rows of at least 256 groups in a range with enough rows to fill the GPU one run per wave:
    // the line-owning form (rowrun_kernels.hip.h), a wave per run of whole rows, every 128-byte line of the text has one
    // writer.  Otherwise (short rows, or a few very long ones): a wave per row segment (rowgen_kernels.hip.h).
    p.nseg = c->G ? (c->G + 64u * c->seg_iters - 1) / (64u * c->seg_iters) : 1;
    p.seg_groups = c->G ? ((c->G + p.nseg - 1) / p.nseg + 63u) / 64u * 64u : 64u;
    uint64_t row_bytes_min = 2ull * c->S; // a Y / MT row's cells; every row is longer
    uint32_t rpw = 1;
    if (c->G >= 256u) {
        // ~64 KiB of text per wave, but never fewer than ~16k waves while the range has the rows for it
        rpw = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(64, 65536ull / (4ull * c->S)));
        while (rpw > 1 && n_rows / rpw < 16384) rpw /= 2;
        if (c->force_rpw) rpw = c->force_rpw;
    }
    bool use_runs = c->G >= 256u && row_bytes_min >= 256 && !c->force_segments && (p.nseg == 1 || n_rows / rpw >= c->min_run_waves);
    // short rows (dense packed pitch): the lanes of a trip span rows (rowmix_kernels.hip.h); a wave takes about 2048 groups
    const bool use_mix = c->S >= 1u && c->Wp == c->W && c->G < c->mix_max_groups && !c->force_segments && !c->no_rowmix;
    if (use_mix) {
        use_runs = false;
        // up to 64 rows (one lane sets up one row), about 3,000 groups; of the candidates the one whose last trip is
        // fullest (a wave's last trip costs a whole trip: 41 rows of 50 groups end in a trip of 2)
        rpw = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(1, (3200 + c->G - 1) / c->G));
        {
            uint32_t best = rpw;
            double best_fill = 0;
            for (uint32_t r = rpw; r >= 1 && r + 16 > rpw; --r) {
                const uint64_t groups = (uint64_t)r * c->G, trips = (groups + 63) / 64;
                const double fill = (double)groups / (64.0 * trips);
                if (fill > best_fill + 1e-9) { best_fill = fill; best = r; }
            }
            rpw = best;
        }
        while (rpw > 1 && n_rows / rpw < 16384) rpw /= 2;
        if (c->force_rpw) rpw = std::min<uint32_t>(c->force_rpw, 64u);
        if (c->prefix_stride) rpw = std::min<uint32_t>(rpw, dnaf::MIX_PFX_BYTES / c->prefix_stride); // the run's fixed columns must fit their LDS cache
    }
    p.rows_per_wave = rpw;
    const uint64_t items = (use_runs || use_mix) ? (n_rows + rpw - 1) / rpw : n_rows * p.nseg;
    const uint64_t blocks = (items + 3) / 4;
    if (blocks > 0x7FFFFFFFull) return fail(c, DNAF_ERR_UNSUPPORTED, "dnaf_generate: %llu workgroups; generate fewer rows per call", (unsigned long long)blocks);
    // one event pair per launch, on the launch stream (pool bounded: later launches go unrecorded)
    const bool rec = c->n_timed < MAX_TIMED;
    if (rec && c->evs.size() < 2 * (c->n_timed + 1)) {
        hipEvent_t a = nullptr, b = nullptr;
        HIP_TRY(c, hipEventCreate(&a));
        c->evs.push_back(a);
        HIP_TRY(c, hipEventCreate(&b));
        c->evs.push_back(b);
    }
    if (rec) HIP_TRY(c, hipEventRecord(c->evs[2 * c->n_timed], c->stream));
    const dim3 grid((uint32_t)blocks), block(256);
    if (use_mix) {
        const bool full4 = c->S % 4u == 0u;
        const uint32_t form = (what == (DNAF_OUT_TEXT | DNAF_OUT_PACKED) ? 0u : what == DNAF_OUT_TEXT ? 4u : 8u) | (full4 ? 2u : 0u) | (p.mix_x_rows ? 1u : 0u);
#define DNAF_MIX_LAUNCH(T, P, F, X) hipLaunchKernelGGL((dnaf::rowmix_kernel<T, P, F, X>), grid, block, 0, c->stream, p)
        switch (form) { // text? packed? whole groups only? X rows in the trips?
        case 0: DNAF_MIX_LAUNCH(true, true, false, false); break;
        case 1: DNAF_MIX_LAUNCH(true, true, false, true); break;
        case 2: DNAF_MIX_LAUNCH(true, true, true, false); break;
        case 3: DNAF_MIX_LAUNCH(true, true, true, true); break;
        case 4: DNAF_MIX_LAUNCH(true, false, false, false); break;
        case 5: DNAF_MIX_LAUNCH(true, false, false, true); break;
        case 6: DNAF_MIX_LAUNCH(true, false, true, false); break;
        case 7: DNAF_MIX_LAUNCH(true, false, true, true); break;
        case 8: DNAF_MIX_LAUNCH(false, true, false, false); break;
        case 9: DNAF_MIX_LAUNCH(false, true, false, true); break;
        case 10: DNAF_MIX_LAUNCH(false, true, true, false); break;
        default: DNAF_MIX_LAUNCH(false, true, true, true); break;
        }
#undef DNAF_MIX_LAUNCH
    } else if (use_runs) {
        // the cap on workgroups per CU pays for long rows (a wave's passes get closer in time); waves that walk several
        // short rows have more per-row latency to cover and want the full occupancy (S = 4,000: 0.61 without, 0.57 with)
        const uint32_t pad = rpw <= 2 ? c->run_lds_pad : 0u;
        if (what == (DNAF_OUT_TEXT | DNAF_OUT_PACKED)) hipLaunchKernelGGL((dnaf::rowrun_kernel<true, true>), grid, block, pad, c->stream, p);
        else if (what == DNAF_OUT_TEXT) hipLaunchKernelGGL((dnaf::rowrun_kernel<true, false>), grid, block, pad, c->stream, p);
        else hipLaunchKernelGGL((dnaf::rowrun_kernel<false, true>), grid, block, pad, c->stream, p);
    } else if (what == (DNAF_OUT_TEXT | DNAF_OUT_PACKED)) hipLaunchKernelGGL((dnaf::rowgen_kernel<true, true>), grid, block, 0, c->stream, p);
    else if (what == DNAF_OUT_TEXT) hipLaunchKernelGGL((dnaf::rowgen_kernel<true, false>), grid, block, 0, c->stream, p);
    else hipLaunchKernelGGL((dnaf::rowgen_kernel<false, true>), grid, block, 0, c->stream, p);
    HIP_TRY(c, hipGetLastError());
    c->last_kernel = use_mix ? "rowmix_kernel" : use_runs ? "rowrun_kernel" : "rowgen_kernel";
    if (rec) {
        HIP_TRY(c, hipEventRecord(c->evs[2 * c->n_timed + 1], c->stream));
        ++c->n_timed;
    }
    ++c->n_launches;
    return DNAF_OK;
}

int dnaf_sync(dnaf_ctx* c)
{
    if (!c) return DNAF_ERR_BAD_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DNAF_OK;
}

int dnaf_last_generate_ms(dnaf_ctx* c, float* ms)
{
    if (!c || !ms) return DNAF_ERR_BAD_ARG;
    *ms = 0.f;
    if (!c->n_timed) return fail(c, DNAF_ERR_STATE, "dnaf_last_generate_ms: no kernel launch was recorded");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventSynchronize(c->evs[2 * c->n_timed - 1]));
    HIP_TRY(c, hipEventElapsedTime(ms, c->evs[2 * c->n_timed - 2], c->evs[2 * c->n_timed - 1]));
    return DNAF_OK;
}

int dnaf_timing_begin(dnaf_ctx* c)
{
    if (!c) return DNAF_ERR_BAD_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->n_timed = 0;
    c->n_launches = 0;
    return DNAF_OK;
}

int dnaf_timing_end(dnaf_ctx* c, double* total_ms, uint64_t* n_recorded, uint64_t* n_launches)
{
    if (!c || !total_ms) return DNAF_ERR_BAD_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double tot = 0.0;
    for (size_t i = 0; i < c->n_timed; ++i) {
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->evs[2 * i], c->evs[2 * i + 1]));
        tot += ms;
    }
    *total_ms = tot;
    if (n_recorded) *n_recorded = c->n_timed;
    if (n_launches) *n_launches = c->n_launches;
    return DNAF_OK;
}

int dnaf_read_text(dnaf_ctx* c, void* dst, uint64_t dst_cap, uint64_t* n_bytes)
{
    if (!c) return DNAF_ERR_BAD_ARG;
    if (!c->have_output || !(c->last_what & DNAF_OUT_TEXT)) return fail(c, DNAF_ERR_STATE, "dnaf_read_text: no text was generated");
    const uint64_t n = c->h_text_off[(size_t)(c->last_begin + c->last_rows)] - c->h_text_off[(size_t)c->last_begin];
    if (n_bytes) *n_bytes = n;
    if (n > dst_cap) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_read_text: need %llu bytes, have %llu", (unsigned long long)n, (unsigned long long)dst_cap);
    if (n && !dst) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_read_text: dst is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    if (n) HIP_TRY(c, hipMemcpyAsync(dst, c->d_text, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DNAF_OK;
}

// Per-row alternative-allele counts of the last generated packed matrix: counts[2*j] among the control
// samples, counts[2*j+1] among the cases (set bits of the 2-bit codes; one wave per row).
int dnaf_allele_counts(dnaf_ctx* c, uint32_t* counts, uint64_t cap_rows)
{
    if (!c) return DNAF_ERR_BAD_ARG;
    if (!c->have_output || !(c->last_what & DNAF_OUT_PACKED)) return fail(c, DNAF_ERR_STATE, "dnaf_allele_counts: no packed matrix was generated");
    const uint64_t n = c->last_rows;
    if (n > cap_rows || (n && !counts)) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_allele_counts: need room for %llu rows", (unsigned long long)n);
    if (n == 0) return DNAF_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (n > c->counts_cap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        dev_free(c->d_counts);
        c->counts_cap = 0;
        HIP_TRY(c, hipMalloc((void**)&c->d_counts, n * 2 * sizeof(uint32_t)));
        c->counts_cap = n;
    }
    hipLaunchKernelGGL(dnaf::allele_count_kernel, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, c->stream,
                       (const uint32_t*)c->d_packed, (const uint32_t*)c->d_ctrl2, c->W, c->Wp, n, c->d_counts);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(counts, c->d_counts, n * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DNAF_OK;
}

// PLINK 1 .bed rows (SNP-major, 2 bits per sample, A1 = ALT, A2 = REF) of the last generated packed matrix.
int dnaf_export_bed(dnaf_ctx* c, uint64_t* n_bytes)
{
    if (!c) return DNAF_ERR_BAD_ARG;
    if (!c->have_output || !(c->last_what & DNAF_OUT_PACKED)) return fail(c, DNAF_ERR_STATE, "dnaf_export_bed: no packed matrix was generated");
    c->have_bed = false;
    const uint64_t row_bytes = ((uint64_t)c->S + 3) / 4, n = c->last_rows, total = n * row_bytes;
    if (n_bytes) *n_bytes = total;
    c->bed_bytes = total;
    if (total == 0) { c->have_bed = true; return DNAF_OK; }
    HIP_TRY(c, hipSetDevice(c->device));
    if (total > c->bed_cap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        dev_free(c->d_bed);
        c->bed_cap = 0;
        HIP_TRY(c, hipMalloc((void**)&c->d_bed, total));
        c->bed_cap = total;
    }
    const uint64_t words = n * c->W;
    hipLaunchKernelGGL(dnaf::bed_kernel, dim3((uint32_t)((words + 255) / 256)), dim3(256), 0, c->stream, (const uint32_t*)c->d_packed,
                       (const RowRec*)(c->d_recs + c->last_begin), (const uint8_t*)c->d_male_nib, c->S, c->W, c->Wp, c->G, n, c->d_bed);
    HIP_TRY(c, hipGetLastError());
    c->have_bed = true;
    return DNAF_OK;
}

int dnaf_read_bed(dnaf_ctx* c, void* dst, uint64_t dst_cap, uint64_t* n_bytes)
{
    if (!c) return DNAF_ERR_BAD_ARG;
    if (!c->have_bed) return fail(c, DNAF_ERR_STATE, "dnaf_read_bed: nothing was exported");
    if (n_bytes) *n_bytes = c->bed_bytes;
    if (c->bed_bytes > dst_cap || (c->bed_bytes && !dst)) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_read_bed: need %llu bytes", (unsigned long long)c->bed_bytes);
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->bed_bytes) HIP_TRY(c, hipMemcpyAsync(dst, c->d_bed, c->bed_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DNAF_OK;
}

int dnaf_read_text_range(dnaf_ctx* c, uint64_t byte_begin, uint64_t n_bytes, void* dst)
{
    if (!c) return DNAF_ERR_BAD_ARG;
    if (!c->have_output || !(c->last_what & DNAF_OUT_TEXT)) return fail(c, DNAF_ERR_STATE, "dnaf_read_text_range: no text was generated");
    const uint64_t n = c->h_text_off[(size_t)(c->last_begin + c->last_rows)] - c->h_text_off[(size_t)c->last_begin];
    if (byte_begin > n || n_bytes > n - byte_begin) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_read_text_range: [%llu,+%llu) outside the %llu bytes generated", (unsigned long long)byte_begin, (unsigned long long)n_bytes, (unsigned long long)n);
    if (n_bytes && !dst) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_read_text_range: dst is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    if (n_bytes) HIP_TRY(c, hipMemcpyAsync(dst, c->d_text + byte_begin, n_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DNAF_OK;
}

int dnaf_read_packed(dnaf_ctx* c, uint32_t* dst, uint64_t dst_cap_words, uint64_t* n_words)
{
    if (!c) return DNAF_ERR_BAD_ARG;
    if (!c->have_output || !(c->last_what & DNAF_OUT_PACKED)) return fail(c, DNAF_ERR_STATE, "dnaf_read_packed: no packed matrix was generated");
    const uint64_t n = c->last_rows * (uint64_t)c->W;
    if (n_words) *n_words = n;
    if (n > dst_cap_words) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_read_packed: need %llu words, have %llu", (unsigned long long)n, (unsigned long long)dst_cap_words);
    if (n && !dst) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_read_packed: dst is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    // the device rows are Wp words apart; the caller gets the dense [n_rows][W] matrix: one plain copy where the pitch is
    // the row length; otherwise the rows are first packed densely on the device and copied in one piece (250,000 rows of
    // 5,000 bytes: 0.06-0.10 s; hipMemcpy2DAsync straight from the pitched rows: 0.08-0.12 s -- profiles/r03_read_packed.txt)
    if (n && c->Wp == c->W) HIP_TRY(c, hipMemcpyAsync(dst, c->d_packed, n * 4, hipMemcpyDeviceToHost, c->stream));
    else if (n) {
        if (n * 4 > c->dense_cap) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            dev_free(c->d_dense);
            c->dense_cap = 0;
            HIP_TRY(c, hipMalloc((void**)&c->d_dense, n * 4));
            c->dense_cap = n * 4;
        }
        const uint64_t total = n;
        hipLaunchKernelGGL(dnaf::packed_compact_kernel, dim3((uint32_t)std::min<uint64_t>((total + 255) / 256, 1u << 20)), dim3(256), 0, c->stream,
                           (const uint32_t*)c->d_packed, c->d_dense, (uint64_t)c->last_rows, c->W, c->Wp);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(dst, c->d_dense, n * 4, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DNAF_OK;
}

#ifdef DNAF_GTZ_PROFILE
// debug builds: read and reset the stage clock of gtz_encode (tools/gtz_stage_profile.py)
int dnaf_debug_gtz_profile(unsigned long long* out16)
{
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(dnaf::gtz_prof), sizeof z) != hipSuccess) return DNAF_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(dnaf::gtz_prof), z, sizeof z) != hipSuccess) return DNAF_ERR_HIP;
    return DNAF_OK;
}
#endif

#ifdef DNAF_MIX_PROFILE
// debug builds: read and reset the stage clock of rowmix_kernel (tools/mix_stage_profile.py)
int dnaf_debug_mix_profile(unsigned long long* out16)
{
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(dnaf::mix_prof), sizeof z) != hipSuccess) return DNAF_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(dnaf::mix_prof), z, sizeof z) != hipSuccess) return DNAF_ERR_HIP;
    return DNAF_OK;
}
#endif

int dnaf_packed_pitch_words(const dnaf_ctx* c, uint64_t* pitch_words)
{
    if (!c || !pitch_words) return DNAF_ERR_BAD_ARG;
    if (!c->have_samples) return fail(const_cast<dnaf_ctx*>(c), DNAF_ERR_STATE, "dnaf_packed_pitch_words: call dnaf_set_samples first");
    *pitch_words = c->Wp;
    return DNAF_OK;
}

int dnaf_device_buffers(dnaf_ctx* c, void** d_text, uint64_t* text_bytes, void** d_packed, uint64_t* packed_words)
{
    if (!c) return DNAF_ERR_BAD_ARG;
    if (!c->have_output) return fail(c, DNAF_ERR_STATE, "dnaf_device_buffers: nothing was generated");
    const bool t = c->last_what & DNAF_OUT_TEXT, k = c->last_what & DNAF_OUT_PACKED;
    if (d_text) *d_text = t ? c->d_text : nullptr;
    if (text_bytes) *text_bytes = t ? c->h_text_off[(size_t)(c->last_begin + c->last_rows)] - c->h_text_off[(size_t)c->last_begin] : 0;
    if (d_packed) *d_packed = k ? c->d_packed : nullptr;
    if (packed_words) *packed_words = k ? c->last_rows * (uint64_t)c->Wp : 0; // rows are dnaf_packed_pitch_words() apart
    return DNAF_OK;
}

// ---- on-GPU BGZF encoder: the sink right after the path (Bio.bgzf.BgzfWriter, pop_factory.py:403, :449) ----
int dnaf_bgzf_compress(dnaf_ctx* c, uint32_t flags, uint64_t* n_bytes)
{
    if (!c) return DNAF_ERR_BAD_ARG;
    if (!c->have_output || !(c->last_what & DNAF_OUT_TEXT)) return fail(c, DNAF_ERR_STATE, "dnaf_bgzf_compress: no text was generated");
    if (flags & ~(DNAF_BGZF_REUSE_TABLES | DNAF_BGZF_TEXT_DOMAIN)) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_bgzf_compress: flags = %u", flags);
    // members that lie in biallelic autosome rows are made from the genotype matrix when it was generated too
    bool gtz = (c->last_what & DNAF_OUT_PACKED) && !(flags & DNAF_BGZF_TEXT_DOMAIN) && c->S >= 2;
    HIP_TRY(c, hipSetDevice(c->device));
    const uint64_t tbytes = c->h_text_off[(size_t)(c->last_begin + c->last_rows)] - c->h_text_off[(size_t)c->last_begin];
    c->have_bgzf = false;
    c->bgzf_bytes = 0;
    c->bgzf_ms = 0.f;
    if (n_bytes) *n_bytes = 0;
    if (tbytes == 0) { c->have_bgzf = true; return DNAF_OK; }
    const uint64_t nb64 = (tbytes + dnaf::BGZF_IN - 1) / dnaf::BGZF_IN;
    if (nb64 > 0x7FFFFFFFull) return fail(c, DNAF_ERR_UNSUPPORTED, "dnaf_bgzf_compress: too many blocks");
    const uint32_t nb = (uint32_t)nb64;
    if (!c->zev[0]) { HIP_TRY(c, hipEventCreate(&c->zev[0])); HIP_TRY(c, hipEventCreate(&c->zev[1])); }
    if (!c->d_tables) HIP_TRY(c, hipMalloc((void**)&c->d_tables, sizeof(dnaf::BgzfTables)));
    if (!c->d_hist) HIP_TRY(c, hipMalloc((void**)&c->d_hist, 513 * sizeof(uint32_t)));
    if (nb > c->blocks_cap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        dev_free(c->d_sizes);
        dev_free(c->d_thread_bits);
        dev_free(c->d_crcs);
        dev_free(c->d_large);
        dev_free(c->d_scan_u64);
        dev_free(c->d_scan_u32);
        dev_free(c->d_offsets);
        dev_free(c->d_member_row);
        dev_free(c->d_fast);
        dev_free(c->d_todo);
        c->blocks_cap = 0;
        hipError_t e = hipMalloc((void**)&c->d_todo, ((uint64_t)nb + 1) * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void**)&c->d_member_row, (uint64_t)nb * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void**)&c->d_fast, (uint64_t)nb);
        if (e == hipSuccess) e = hipMalloc((void**)&c->d_crcs, (uint64_t)nb * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void**)&c->d_large, ((uint64_t)nb + 1) * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void**)&c->d_sizes, (uint64_t)nb * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void**)&c->d_thread_bits, (uint64_t)nb * dnaf::BGZF_THREADS * sizeof(uint16_t));
        if (e == hipSuccess) e = hipMalloc((void**)&c->d_offsets, ((uint64_t)nb + 1) * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc((void**)&c->d_scan_u64, ((uint64_t)nb / dnaf::SCAN_BLOCK + 1) * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc((void**)&c->d_scan_u32, ((uint64_t)nb / dnaf::SCAN_BLOCK + 1) * sizeof(uint32_t));
        if (e != hipSuccess) { // all or nothing: a partial set is not kept
            dev_free(c->d_scan_u64); dev_free(c->d_scan_u32);
            dev_free(c->d_todo); dev_free(c->d_member_row); dev_free(c->d_fast); dev_free(c->d_crcs);
            dev_free(c->d_large); dev_free(c->d_sizes); dev_free(c->d_thread_bits); dev_free(c->d_offsets);
            return fail(c, e == hipErrorOutOfMemory ? DNAF_ERR_OOM : DNAF_ERR_HIP, "dnaf_bgzf_compress: %s", hipGetErrorString(e));
        }
        c->blocks_cap = nb;
    }
    const uint32_t scan_blocks = (nb + dnaf::SCAN_BLOCK - 1) / dnaf::SCAN_BLOCK;
    dnaf::BgzfParams p;
    p.text = c->d_text;
    p.n_bytes = tbytes;
    p.n_blocks = nb;
    p.sample_step = nb > 512 ? nb / 512 : 1;
    p.tables = c->d_tables;
    p.hist = c->d_hist;
    p.sizes = c->d_sizes;
    p.thread_bits = c->d_thread_bits;
    p.crcs = c->d_crcs;
    p.large_list = c->d_large;
    p.offsets = c->d_offsets;
    p.out = nullptr;
    if (gtz && (uint64_t)nb * dnaf::GTZ_STRIDE > c->stage_cap) {
        // staging slots of the genotype-domain members; without room for them every member is made from the text
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        dev_free(c->d_stage);
        c->stage_cap = 0;
        if (hipMalloc((void**)&c->d_stage, (uint64_t)nb * dnaf::GTZ_STRIDE) == hipSuccess) c->stage_cap = (uint64_t)nb * dnaf::GTZ_STRIDE;
        else { (void)hipGetLastError(); c->d_stage = nullptr; gtz = false; }
    }
    p.fast = gtz ? c->d_fast : nullptr;
    p.todo = nullptr;
    uint32_t n_todo = nb; // members the text-domain kernels encode
    float ms_total = 0.f, ms = 0.f;
    dnaf::GtzParams g;
    if (gtz) {
        if (!c->d_gtz_tables) HIP_TRY(c, hipMalloc((void**)&c->d_gtz_tables, sizeof(dnaf::GtzTables)));
        if (!c->d_gtz_hist) HIP_TRY(c, hipMalloc((void**)&c->d_gtz_hist, dnaf::GTZ_NCLASS * dnaf::GTZ_NSYM * sizeof(uint32_t)));
        g.recs = c->d_recs;
        g.prefix = c->d_prefix;
        g.packed = (const uint32_t*)c->d_packed;
        g.text_base = c->h_text_off[(size_t)c->last_begin];
        g.row_begin = c->last_begin;
        g.n_rows = c->last_rows;
        g.n_bytes = tbytes;
        g.n_blocks = nb;
        g.S = (uint32_t)c->S;
        g.W = c->W;
        g.Wp = c->Wp;
        g.sample_step = nb > 2048 ? nb / 2048 : 1;
        g.tables = c->d_gtz_tables;
        g.member_row = c->d_member_row;
        g.hist = c->d_gtz_hist;
        g.sizes = c->d_sizes;
        g.fast = c->d_fast;
        g.stage = c->d_stage;
        g.offsets = c->d_offsets;
        g.out = nullptr;
        HIP_TRY(c, hipEventRecord(c->zev[0], c->stream));
        hipLaunchKernelGGL(dnaf::gtz_locate, dim3((nb + 255) / 256), dim3(256), 0, c->stream, g);
        HIP_TRY(c, hipGetLastError());
        // one item (64-cell chunk or literal span) per thread: a member of long rows has at most 241 + 3 x (rows it
        // overlaps) of them; where rows are short enough for more than 256, the 512-thread form runs
        const bool many = 241u + 3u * (dnaf::BGZF_IN / (4u * (g.S ? g.S : 1u)) + 2u) > dnaf::GTZ_THREADS;
        if (!(c->have_gtz_tables && (flags & DNAF_BGZF_REUSE_TABLES))) {
            static thread_local uint32_t h[dnaf::GTZ_NCLASS * dnaf::GTZ_NSYM];
            HIP_TRY(c, hipMemsetAsync(c->d_gtz_hist, 0, sizeof h, c->stream));
            if (many) hipLaunchKernelGGL((dnaf::gtz_encode<true, dnaf::GTZ_THREADS_MANY>), dim3((nb + g.sample_step - 1) / g.sample_step), dim3(dnaf::GTZ_THREADS_MANY), 0, c->stream, g);
            else hipLaunchKernelGGL((dnaf::gtz_encode<true, dnaf::GTZ_THREADS>), dim3((nb + g.sample_step - 1) / g.sample_step), dim3(dnaf::GTZ_THREADS), 0, c->stream, g);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipMemcpyAsync(h, c->d_gtz_hist, sizeof h, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            // the histogram kernel's self-check of the encoder's bit count (spare slot of every class, gtz_kernels.hip.h)
            for (uint32_t k = 0; k < dnaf::GTZ_NCLASS; ++k)
                if (h[k * dnaf::GTZ_NSYM + dnaf::GTZ_NSYM - 1])
                    return fail(c, DNAF_ERR_STATE, "dnaf_bgzf_compress: the encoder's bit count disagrees with its token walk on %u sampled chunks (class %u): "
                                "refusing to write members", h[k * dnaf::GTZ_NSYM + dnaf::GTZ_NSYM - 1], k);
            static thread_local dnaf::GtzTables gt;
            if (!dnaf::gtz_build_tables(h, &gt)) return fail(c, DNAF_ERR_UNSUPPORTED, "dnaf_bgzf_compress: block header does not fit");
            HIP_TRY(c, hipMemcpy(c->d_gtz_tables, &gt, sizeof gt, hipMemcpyHostToDevice));
            c->have_gtz_tables = true;
        }
        if (many) hipLaunchKernelGGL((dnaf::gtz_encode<false, dnaf::GTZ_THREADS_MANY>), dim3(nb), dim3(dnaf::GTZ_THREADS_MANY), 0, c->stream, g);
        else hipLaunchKernelGGL((dnaf::gtz_encode<false, dnaf::GTZ_THREADS>), dim3(nb), dim3(dnaf::GTZ_THREADS), 0, c->stream, g);
        HIP_TRY(c, hipGetLastError());
        hipLaunchKernelGGL(dnaf::bgzf_unset_totals, dim3(scan_blocks), dim3(dnaf::SCAN_THREADS), 0, c->stream, (const uint8_t*)c->d_fast, nb, c->d_scan_u32);
        hipLaunchKernelGGL(dnaf::bgzf_unset_apply, dim3(scan_blocks), dim3(dnaf::SCAN_THREADS), 0, c->stream, (const uint8_t*)c->d_fast, nb, (const uint32_t*)c->d_scan_u32, c->d_todo);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(c->zev[1], c->stream));
        HIP_TRY(c, hipMemcpyAsync(&n_todo, c->d_todo, sizeof n_todo, hipMemcpyDeviceToHost, c->stream));
        p.todo = c->d_todo;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipEventElapsedTime(&ms, c->zev[0], c->zev[1]));
        ms_total += ms;
    }
    if (!(c->have_tables && (flags & DNAF_BGZF_REUSE_TABLES))) {
        // 1. token histogram over a sample of blocks -> one Huffman table for the whole range
        uint32_t h[513];
        HIP_TRY(c, hipMemsetAsync(c->d_hist, 0, sizeof h, c->stream));
        HIP_TRY(c, hipEventRecord(c->zev[0], c->stream));
        hipLaunchKernelGGL(dnaf::bgzf_hist, dim3((nb + p.sample_step - 1) / p.sample_step), dim3(dnaf::BGZF_THREADS), 0, c->stream, p);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(c->zev[1], c->stream));
        HIP_TRY(c, hipMemcpyAsync(h, c->d_hist, sizeof h, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipEventElapsedTime(&ms, c->zev[0], c->zev[1]));
        ms_total += ms;
        uint64_t lit[256], mat[256];
        for (int i = 0; i < 256; ++i) { lit[i] = h[i]; mat[i] = h[256 + i]; }
        static thread_local dnaf::BgzfTables tables;
        dnaf::bgzf_build_tables(lit, mat, h[512], &tables);
        HIP_TRY(c, hipMemcpy(c->d_tables, &tables, sizeof tables, hipMemcpyHostToDevice));
        c->have_tables = true;
    }
    // 2. exact member sizes, 3. their exclusive scan
    uint64_t total = 0;
    HIP_TRY(c, hipEventRecord(c->zev[0], c->stream));
    if (n_todo) {
        hipLaunchKernelGGL(dnaf::bgzf_size, dim3(n_todo), dim3(dnaf::BGZF_THREADS), 0, c->stream, p);
        HIP_TRY(c, hipGetLastError());
    }
    uint32_t n_large = 0;
    hipLaunchKernelGGL(dnaf::bgzf_scan_totals, dim3(scan_blocks), dim3(dnaf::SCAN_THREADS), 0, c->stream, (const uint32_t*)c->d_sizes, nb, c->d_scan_u64, c->d_scan_u32);
    hipLaunchKernelGGL(dnaf::bgzf_scan_apply, dim3(scan_blocks), dim3(dnaf::SCAN_THREADS), 0, c->stream, (const uint32_t*)c->d_sizes, nb, (const uint64_t*)c->d_scan_u64,
                       (const uint32_t*)c->d_scan_u32, c->d_offsets, c->d_large);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->zev[1], c->stream));
    HIP_TRY(c, hipMemcpyAsync(&total, c->d_offsets + nb, sizeof total, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&n_large, c->d_large, sizeof n_large, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipEventElapsedTime(&ms, c->zev[0], c->zev[1]));
    ms_total += ms;
    if (total > c->bgzf_cap) {
        dev_free(c->d_bgzf);
        c->bgzf_cap = 0;
        const uint64_t want = total + total / 4 + 65536; // headroom: later tiles compress to similar sizes
        HIP_TRY(c, hipMalloc((void**)&c->d_bgzf, want));
        c->bgzf_cap = want;
    }
    // 4. encode every member at its offset
    p.out = c->d_bgzf;
    HIP_TRY(c, hipEventRecord(c->zev[0], c->stream));
    if (n_todo) {
        hipLaunchKernelGGL((dnaf::bgzf_encode<dnaf::BGZF_OUT_SMALL>), dim3(n_todo), dim3(dnaf::BGZF_THREADS), 0, c->stream, p);
        HIP_TRY(c, hipGetLastError());
    }
    if (n_large) {
        hipLaunchKernelGGL((dnaf::bgzf_encode<dnaf::BGZF_OUT_LARGE>), dim3(n_large), dim3(dnaf::BGZF_THREADS), 0, c->stream, p);
        HIP_TRY(c, hipGetLastError());
    }
    if (gtz) {
        g.out = c->d_bgzf;
        hipLaunchKernelGGL(dnaf::gtz_compact, dim3(nb), dim3(256), 0, c->stream, g);
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipEventRecord(c->zev[1], c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipEventElapsedTime(&ms, c->zev[0], c->zev[1]));
    ms_total += ms;
    c->bgzf_ms = ms_total;
    c->bgzf_bytes = total;
    c->have_bgzf = true;
    if (n_bytes) *n_bytes = total;
    return DNAF_OK;
}

int dnaf_read_bgzf(dnaf_ctx* c, void* dst, uint64_t dst_cap, uint64_t* n_bytes)
{
    if (!c) return DNAF_ERR_BAD_ARG;
    if (!c->have_bgzf) return fail(c, DNAF_ERR_STATE, "dnaf_read_bgzf: nothing was compressed");
    if (n_bytes) *n_bytes = c->bgzf_bytes;
    if (c->bgzf_bytes > dst_cap) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_read_bgzf: need %llu bytes, have %llu", (unsigned long long)c->bgzf_bytes, (unsigned long long)dst_cap);
    if (c->bgzf_bytes && !dst) return fail(c, DNAF_ERR_BAD_ARG, "dnaf_read_bgzf: dst is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->bgzf_bytes) HIP_TRY(c, hipMemcpyAsync(dst, c->d_bgzf, c->bgzf_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DNAF_OK;
}

// Copy `n` bytes from device memory into the context's pinned staging buffer (grown on demand).
static int fetch_pinned(dnaf_ctx* c, const void* d_src, uint64_t n, const void** host)
{
    HIP_TRY(c, hipSetDevice(c->device));
    const int k = c->pinned_next;
    c->pinned_next = (k + 1) % dnaf_ctx::N_PINNED;
    if (n > c->pinned_cap[k]) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->h_pinned[k]) (void)hipHostFree(c->h_pinned[k]);
        c->h_pinned[k] = nullptr;
        c->pinned_cap[k] = 0;
        const uint64_t want = n + n / 4 + 4096;
        HIP_TRY(c, hipHostMalloc(&c->h_pinned[k], want, hipHostMallocDefault));
        c->pinned_cap[k] = want;
    }
    if (n) HIP_TRY(c, hipMemcpyAsync(c->h_pinned[k], d_src, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *host = c->h_pinned[k];
    return DNAF_OK;
}

int dnaf_fetch_bgzf(dnaf_ctx* c, const void** host, uint64_t* n_bytes)
{
    if (!c || !host) return DNAF_ERR_BAD_ARG;
    if (!c->have_bgzf) return fail(c, DNAF_ERR_STATE, "dnaf_fetch_bgzf: nothing was compressed");
    if (n_bytes) *n_bytes = c->bgzf_bytes;
    return fetch_pinned(c, c->d_bgzf, c->bgzf_bytes, host);
}

int dnaf_fetch_text(dnaf_ctx* c, const void** host, uint64_t* n_bytes)
{
    if (!c || !host) return DNAF_ERR_BAD_ARG;
    if (!c->have_output || !(c->last_what & DNAF_OUT_TEXT)) return fail(c, DNAF_ERR_STATE, "dnaf_fetch_text: no text was generated");
    const uint64_t n = c->h_text_off[(size_t)(c->last_begin + c->last_rows)] - c->h_text_off[(size_t)c->last_begin];
    if (n_bytes) *n_bytes = n;
    return fetch_pinned(c, c->d_text, n, host);
}

int dnaf_last_bgzf_ms(dnaf_ctx* c, float* ms)
{
    if (!c || !ms) return DNAF_ERR_BAD_ARG;
    if (!c->have_bgzf) return fail(c, DNAF_ERR_STATE, "dnaf_last_bgzf_ms: nothing was compressed");
    *ms = c->bgzf_ms;
    return DNAF_OK;
}

// Host-only: the tables the encoder would use for a given token histogram (lets CPU tests inflate a
// stream assembled from them).  out must hold sizeof(BgzfTables) bytes; *size receives that size.
int dnaf_bgzf_tables_from_histogram(const uint64_t* lit_hist, const uint64_t* match_hist, uint64_t n_eob, void* out, uint64_t out_cap, uint64_t* size)
{
    if (size) *size = sizeof(dnaf::BgzfTables);
    if (!lit_hist || !match_hist || !out || out_cap < sizeof(dnaf::BgzfTables)) return DNAF_ERR_BAD_ARG;
    dnaf::bgzf_build_tables(lit_hist, match_hist, n_eob, (dnaf::BgzfTables*)out);
    return DNAF_OK;
}

// Host-only: the genotype-domain encoder's tables for hist[8][320] (per class: 286 literal/length symbol counts, then
// 30 distance symbol counts).  out must hold sizeof(GtzTables) bytes; *size receives that size.
int dnaf_gtz_tables_from_histogram(const uint32_t* hist, void* out, uint64_t out_cap, uint64_t* size)
{
    if (size) *size = sizeof(dnaf::GtzTables);
    if (!hist || !out || out_cap < sizeof(dnaf::GtzTables)) return DNAF_ERR_BAD_ARG;
    return dnaf::gtz_build_tables(hist, (dnaf::GtzTables*)out) ? DNAF_OK : DNAF_ERR_UNSUPPORTED;
}

// ---- host helper: fixed columns of a VCF body line, pop_factory.py:503-507 ----
static inline size_t put_i64(uint8_t* p, int64_t v)
{
    char tmp[24];
    size_t n = 0;
    uint64_t u = v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    size_t k = 0;
    if (v < 0) p[k++] = '-';
    while (n) p[k++] = (uint8_t)tmp[--n];
    return k;
}
static inline size_t field_len(const char* f, uint32_t w)
{
    size_t n = 0;
    while (n < w && f[n]) ++n;
    return n;
}

int dnaf_format_prefixes(uint64_t n, const char* chrom, uint32_t chrom_w, const int64_t* pos, const int64_t* id,
                         const char* ref, uint32_t ref_w, const char* alt, uint32_t alt_w, uint8_t* out,
                         uint64_t out_cap, uint64_t* off)
{
    if (!off || (n && (!chrom || !pos || !id || !ref || !alt))) return DNAF_ERR_BAD_ARG;
    static const char tail[] = "\t40\tPASS\t.\tGT\t";
    const size_t tail_n = sizeof tail - 1;
    uint64_t o = 0;
    uint8_t scratch[64];
    for (uint64_t j = 0; j < n; ++j) {
        const char *c = chrom + j * chrom_w, *r = ref + j * ref_w, *a = alt + j * alt_w;
        const size_t cl = field_len(c, chrom_w), rl = field_len(r, ref_w), al = field_len(a, alt_w);
        off[j] = o;
        const size_t pl = put_i64(scratch, pos[j]), il = put_i64(scratch + 24, id[j]);
        const uint64_t need = cl + 1 + pl + 1 + 2 + il + 1 + rl + 1 + al + tail_n;
        if (out) {
            if (o + need > out_cap) return DNAF_ERR_BAD_ARG;
            uint8_t* p = out + o;
            memcpy(p, c, cl); p += cl; *p++ = '\t';
            memcpy(p, scratch, pl); p += pl; *p++ = '\t';
            *p++ = 'r'; *p++ = 's';
            memcpy(p, scratch + 24, il); p += il; *p++ = '\t';
            memcpy(p, r, rl); p += rl; *p++ = '\t';
            memcpy(p, a, al); p += al;
            memcpy(p, tail, tail_n);
        }
        o += need;
    }
    off[n] = o;
    return DNAF_OK;
}

} // extern "C"

// ---- numpy-compatible MT19937 mode: host-side jump helpers (also used by CPU tests) ----
extern "C" {

// window[624] = the 624 untempered words that start J >= 1 words after base[624] in the MT19937 word
// sequence (host evaluation, O(19937 * 624); tests and one-off jumps only).
int dnaf_mt_jump_host(const uint32_t* base, uint64_t J, uint32_t* window)
{
    if (!base || !window || J == 0) return DNAF_ERR_BAD_ARG;
    dnaf::MtPoly g;
    mt_jumper().xpow_mod(J - 1, &g);
    dnaf::mt_jump_window_host(base, g, window);
    return DNAF_OK;
}

// numpy.random.choice(pop_size, size, replace=False) of the legacy RandomState (the reference's deleterious picks,
// pop_factory.py:576): permutation(pop_size)[:size], i.e. numpy's Fisher-Yates shuffle of arange(pop_size) -- for i =
// pop_size-1 .. 1: j = random_interval(i) (32-bit draws, masked rejection), swap(x[i], x[j]) -- of which only the first
// `size` entries are wanted.  The draws are replayed exactly (the generator state key[624] / *pos moves on as numpy's
// would), but nothing is shuffled: the j's are kept and each wanted position is traced back through the swaps in reverse
// order, one sequential pass.  2 s -> 0.25 s per pick at 40 M SNPs (BASELINE config 4 picks 14 times).
int dnaf_numpy_legacy_choice(uint32_t* key, uint32_t* pos, uint64_t pop_size, uint32_t size, uint64_t* out_idx)
{
    if (!key || !pos || *pos > 624u || (size && !out_idx)) return DNAF_ERR_BAD_ARG;
    if (size > pop_size || pop_size > 0xFFFFFFFFull) return DNAF_ERR_UNSUPPORTED; // numpy raises / would use 64-bit draws
    if (pop_size == 0) return DNAF_OK;
    std::vector<uint32_t> js;
    try {
        js.resize((size_t)pop_size);
    } catch (const std::bad_alloc&) {
        return DNAF_ERR_OOM;
    }
    // tempered outputs a block of 624 at a time (plain loops the compiler vectorises), consumed in order
    uint32_t p = *pos;
    uint32_t out[624];
    auto temper_block = [&]() {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = key[i];
            y ^= y >> 11;
            y ^= (y << 7) & 0x9D2C5680u;
            y ^= (y << 15) & 0xEFC60000u;
            y ^= y >> 18;
            out[i] = y;
        }
    };
    auto twist = [&]() {
        auto mix = [](uint32_t a, uint32_t b, uint32_t c) { const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu); return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); };
        int i = 0;
        for (; i < 624 - 397; ++i) key[i] = mix(key[i], key[i + 1], key[i + 397]);
        for (; i < 623; ++i) key[i] = mix(key[i], key[i + 1], key[i + 397 - 624]);
        key[623] = mix(key[623], key[0], key[396]);
    };
    if (p < 624u) temper_block();
    uint32_t mask = (uint32_t)(pop_size - 1); // smallest 2^k - 1 >= i, kept as i falls
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    for (uint64_t i = pop_size - 1; i >= 1; --i) {
        if ((uint32_t)i <= (mask >> 1)) mask >>= 1;
        uint32_t v;
        do {
            if (p >= 624u) { twist(); temper_block(); p = 0; }
            v = out[p++] & mask;
        } while (v > (uint32_t)i);
        js[(size_t)i] = v;
    }
    *pos = p;
    // x_final[t] for t < size: undo the swaps from the last executed (i = 1) to the first (i = pop_size-1)
    for (uint32_t t = 0; t < size; ++t) out_idx[t] = t;
    // (a swap (i, j) matters only if a traced position is i or j; i is rarely one -- positions start below `size` and move up
    // only through such swaps -- so the test is a handful of compares per swap)
    for (uint64_t i = 1; i < pop_size; ++i) {
        const uint64_t j = js[(size_t)i];
        bool hit = false;
        for (uint32_t t = 0; t < size; ++t) hit |= (out_idx[t] == i) | (out_idx[t] == j);
        if (!hit) continue;
        for (uint32_t t = 0; t < size; ++t) {
            const uint64_t at = out_idx[t];
            out_idx[t] = at == i ? j : (at == j ? i : at);
        }
    }
    return DNAF_OK;
}

// degree and coefficient words (312 x u64) of the characteristic polynomial (tests)
int dnaf_mt_characteristic_polynomial(uint64_t* words312, int* degree)
{
    if (!words312 || !degree) return DNAF_ERR_BAD_ARG;
    const dnaf::MtPoly& phi = mt_jumper().phi;
    memcpy(words312, phi.w, sizeof phi.w);
    int d = -1;
    for (int i = dnaf::MT_POLY_WORDS * 64 - 1; i >= 0 && d < 0; --i)
        if ((phi.w[i / 64] >> (i % 64)) & 1ull) d = i;
    *degree = d;
    return DNAF_OK;
}

} // extern "C"

// ---- host helper: snps.json.gz lines, str(SNPTuples) of pop_factory.py:118-124 ----
extern "C" {

// One JSON object per row, newline-terminated:
//   {"id": <id>, "chromosome": "<chrom>", "position": <pos>, "tuples": {"<ref>": <cum0>, "<alt1>": <cum1>, ...}}
// ref / alt are the fixed-width fields of dnaf_format_prefixes (alt comma-joined); the cumulative
// frequencies are given as indices into a table of already formatted numbers (repr(float), done by the
// caller once per distinct value): num_off[k] .. num_off[k+1] in num_blob.  Pass out == NULL to size.
int dnaf_format_snps_json(uint64_t n, const int64_t* id, const char* chrom, uint32_t chrom_w, const int64_t* pos,
                          const char* ref, uint32_t ref_w, const char* alt, uint32_t alt_w, const uint8_t* n_alleles,
                          const int32_t* cum_idx /* [n][4] */, const char* num_blob, const uint64_t* num_off,
                          uint8_t* out, uint64_t out_cap, uint64_t* n_bytes)
{
    if (!n_bytes || (n && (!id || !chrom || !pos || !ref || !alt || !n_alleles || !cum_idx || !num_blob || !num_off))) return DNAF_ERR_BAD_ARG;
    uint64_t o = 0;
    uint8_t scratch[48];
    auto put = [&](const void* p, size_t len) {
        if (out && o + len <= out_cap) memcpy(out + o, p, len);
        o += len;
    };
    auto put_s = [&](const char* s) { put(s, strlen(s)); };
    for (uint64_t j = 0; j < n; ++j) {
        const char *c = chrom + j * chrom_w, *r = ref + j * ref_w, *a = alt + j * alt_w;
        const size_t cl = field_len(c, chrom_w), rl = field_len(r, ref_w), al = field_len(a, alt_w);
        put_s("{\"id\": ");
        put(scratch, put_i64(scratch, id[j]));
        put_s(", \"chromosome\": \"");
        put(c, cl);
        put_s("\", \"position\": ");
        put(scratch, put_i64(scratch, pos[j]));
        put_s(", \"tuples\": {\"");
        put(r, rl);
        put_s("\": ");
        const int32_t k0 = cum_idx[j * 4];
        put(num_blob + num_off[k0], num_off[k0 + 1] - num_off[k0]);
        size_t p0 = 0;
        for (uint32_t k = 1; k < n_alleles[j]; ++k) {
            size_t p1 = p0;
            while (p1 < al && a[p1] != ',') ++p1;
            put_s(", \"");
            put(a + p0, p1 - p0);
            put_s("\": ");
            const int32_t kk = cum_idx[j * 4 + k];
            put(num_blob + num_off[kk], num_off[kk + 1] - num_off[kk]);
            p0 = p1 < al ? p1 + 1 : p1;
        }
        put_s("}}\n");
    }
    *n_bytes = o;
    if (out && o > out_cap) return DNAF_ERR_BAD_ARG;
    return DNAF_OK;
}

} // extern "C"

// ---- host helper: parse snps.json.gz text (SNPTuples.from_json, pop_factory.py:126-133, :264-272) ----
extern "C" {

// Parses lines of the form written by str(SNPTuples):
//   {"id": 5, "chromosome": "X", "position": 10, "tuples": {"A": 0.1, "T": 1.0}}
// (key order as shown; the key order of "tuples" is the allele order).  alleles: [max_rows][4] fields of
// allele_w bytes, NUL padded.  Returns DNAF_ERR_UNSUPPORTED on any line it does not understand (escapes,
// other key order, more than 4 alleles, over-long fields): the caller then falls back to a JSON parser.
int dnaf_parse_snps_json(const char* text, uint64_t n_bytes, uint64_t max_rows, int64_t* id, char* chrom, uint32_t chrom_w,
                         int64_t* pos, char* alleles, uint32_t allele_w, double* cum, uint8_t* n_alleles, uint64_t* n_rows)
{
    if (!text || !id || !chrom || !pos || !alleles || !cum || !n_alleles || !n_rows) return DNAF_ERR_BAD_ARG;
    const char *p = text, *end = text + n_bytes;
    uint64_t j = 0;
    auto lit = [&](const char* s) {
        const size_t n = strlen(s);
        if ((size_t)(end - p) < n || memcmp(p, s, n) != 0) return false;
        p += n;
        return true;
    };
    auto integer = [&](int64_t* v) {
        bool neg = false;
        if (p < end && *p == '-') { neg = true; ++p; }
        if (p >= end || *p < '0' || *p > '9') return false;
        uint64_t u = 0;
        while (p < end && *p >= '0' && *p <= '9') u = u * 10 + (uint64_t)(*p++ - '0');
        *v = neg ? -(int64_t)u : (int64_t)u;
        return true;
    };
    auto str = [&](char* dst, uint32_t w) { // after the opening quote; stops at the closing quote
        uint32_t k = 0;
        while (p < end && *p != '"') {
            if (*p == '\\' || (unsigned char)*p < 0x20 || k >= w) return false;
            dst[k++] = *p++;
        }
        if (p >= end) return false;
        ++p;
        return true;
    };
    while (p < end) {
        if (*p == '\n') { ++p; continue; }
        if (j >= max_rows) return DNAF_ERR_BAD_ARG;
        memset(chrom + j * chrom_w, 0, chrom_w);
        memset(alleles + j * 4 * allele_w, 0, 4 * (size_t)allele_w);
        if (!lit("{\"id\": ") || !integer(&id[j]) || !lit(", \"chromosome\": \"") || !str(chrom + j * chrom_w, chrom_w) ||
            !lit(", \"position\": ") || !integer(&pos[j]))
            return DNAF_ERR_UNSUPPORTED;
        uint32_t na = 0;
        if (lit(", \"tuples\": {")) {
            while (true) {
                if (na >= 4 || !lit("\"") || !str(alleles + (j * 4 + na) * allele_w, allele_w) || !lit(": ")) return DNAF_ERR_UNSUPPORTED;
                // the number token is copied out first: the text need not be NUL-terminated
                char num[48];
                size_t nl = 0;
                while (p + nl < end && nl + 1 < sizeof num && (isdigit((unsigned char)p[nl]) || p[nl] == '.' || p[nl] == '-' || p[nl] == '+' || p[nl] == 'e' || p[nl] == 'E')) ++nl;
                if (nl == 0 || nl + 1 >= sizeof num) return DNAF_ERR_UNSUPPORTED;
                memcpy(num, p, nl);
                num[nl] = 0;
                char* stop = nullptr;
                cum[j * 4 + na] = strtod(num, &stop);
                if (stop != num + nl) return DNAF_ERR_UNSUPPORTED;
                p += nl;
                // a key that repeats: json.loads (the reference's reader, pop_factory.py:126-133) keeps only the last one
                for (uint32_t k = 0; k < na; ++k)
                    if (!memcmp(alleles + (j * 4 + k) * allele_w, alleles + (j * 4 + na) * allele_w, allele_w)) return DNAF_ERR_UNSUPPORTED;
                ++na;
                if (lit(", ")) continue;
                if (lit("}")) break;
                return DNAF_ERR_UNSUPPORTED;
            }
        }
        if (!lit("}")) return DNAF_ERR_UNSUPPORTED;
        if (p < end && *p != '\n') return DNAF_ERR_UNSUPPORTED;
        n_alleles[j] = (uint8_t)na;
        ++j;
    }
    *n_rows = j;
    return DNAF_OK;
}

} // extern "C"
