/*
 * rowgen_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
 *
 * Scalar CPU restatement of dna-factory's per-SNP row-generation worker
 * (reference: pop_factory.py:471-513 `PopulationFactory.queue_vcf_snps`),
 * used as the checker by tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg.  Nothing under dna-factory_amd/ may import, link or
 * call this file.
 *
 * Parity status: PINNED.  oracle/make_golden.py drives the real reference
 * worker (imported from /root/reference with a stub Bio.bgzf, which the hot
 * path never touches) and the fixtures it wrote under tests/golden/ are
 * checked against this file by tests/test_oracle_golden.py, in both RNG
 * modes:
 *   - ORACLE_RNG_MT19937: the unmodified reference after numpy.random.seed(k)
 *   - ORACLE_RNG_PHILOX : the reference with numpy.random.rand replaced by
 *                         the Philox4x32-10 draw defined in DESIGN.md
 *
 * Every function cites the reference lines it restates.  Comparisons are
 * done in float64 exactly like the reference (the product uses integer
 * thresholds; that derivation is what the parity tests check).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_RNG_PHILOX 0
#define ORACLE_RNG_MT19937 1

#define ORACLE_MAX_ALLELES 8

/* One SNPTuples record (pop_factory.py:74-133) flattened for C. */
typedef struct {
    char chrom[8];                  /* snp.chromosome, e.g. "1", "X", "MT"          */
    int64_t position;               /* snp.position                                  */
    int64_t id;                     /* snp.id (printed as rs%s)                      */
    char ref[16];                   /* ref_allele_tuple()[0]   pop_factory.py:104-109 */
    char alt[64];                   /* alt_alleles()           pop_factory.py:111-116 */
    int32_t n_alleles;              /* len(snp.tuples)                               */
    int32_t _pad;
    double cum[ORACLE_MAX_ALLELES]; /* tuples[k][1], cumulative frequencies          */
} oracle_snp;

/* ------------------------------------------------------------------ */
/* RNG 1: Philox4x32-10 (Salmon et al. 2011; constants as in           */
/* /opt/rocm/include/rocrand/rocrand_philox4x32_10.h).                  */
/* Draw n of SNP row r is word (n & 3) of                               */
/*   philox(counter = (n >> 2, 0, r_lo, r_hi), key = (seed_lo, seed_hi)) */
/* i.e. rocRAND's sequence for (seed, subsequence = r, offset = n).     */
/* ------------------------------------------------------------------ */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

/* exported so tests can pin the block function on known-answer vectors */
void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    philox4x32_10(c, key[0], key[1]);
    memcpy(out, c, sizeof c);
}

/* The uniforms of one row: stands in for numpy.random.rand(2*S), pop_factory.py:477 */
static void philox_row_uniforms(uint64_t seed, uint64_t row, uint64_t n, double* out)
{
    for (uint64_t j = 0; j < n; j += 4) {
        uint32_t c[4] = {(uint32_t)(j >> 2), (uint32_t)((j >> 2) >> 32), (uint32_t)row, (uint32_t)(row >> 32)};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        for (uint64_t w = 0; w < 4 && j + w < n; ++w)
            out[j + w] = (double)c[w] * (1.0 / 4294967296.0); /* exact: u * 2^-32 */
    }
}

/* ------------------------------------------------------------------ */
/* RNG 2: numpy legacy RandomState = MT19937.               */
/* seed: init_genrand; double: (a>>5, b>>6) -> (a*2^26+b)/2^53.         */
/* ------------------------------------------------------------------ */
typedef struct { uint32_t key[624]; int pos; } mt_state;

static void mt_seed(mt_state* s, uint32_t seed)
{
    for (int i = 0; i < 624; ++i) {
        s->key[i] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
    }
    s->pos = 624;
}

static void mt_regen(mt_state* s)
{
    uint32_t* k = s->key;
    for (int i = 0; i < 624; ++i) {
        uint32_t y = (k[i] & 0x80000000u) | (k[(i + 1) % 624] & 0x7fffffffu);
        k[i] = k[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    s->pos = 0;
}

static uint32_t mt_next(mt_state* s)
{
    if (s->pos >= 624) mt_regen(s);
    uint32_t y = s->key[s->pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

static double mt_double(mt_state* s)
{
    uint32_t a = mt_next(s) >> 5, b = mt_next(s) >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

/* exported: numpy.random.seed(seed); numpy.random.rand(n), after skipping `skip` doubles */
void oracle_mt_rand(uint32_t seed, uint64_t skip, uint64_t n, double* out)
{
    mt_state s;
    mt_seed(&s, seed);
    for (uint64_t i = 0; i < skip; ++i) (void)mt_double(&s);
    for (uint64_t i = 0; i < n; ++i) out[i] = mt_double(&s);
}

/* exported: continue from an explicit numpy state (get_state()[1], [2]) */
void oracle_mt_rand_from_state(const uint32_t key[624], int pos, uint64_t skip, uint64_t n, double* out)
{
    mt_state s;
    memcpy(s.key, key, sizeof s.key);
    s.pos = pos;
    for (uint64_t i = 0; i < skip; ++i) (void)mt_double(&s);
    for (uint64_t i = 0; i < n; ++i) out[i] = mt_double(&s);
}

/* ------------------------------------------------------------------ */
/* Worker logic                                                         */
/* ------------------------------------------------------------------ */

/* SNPTuples.pick_allele_index, pop_factory.py:92-95: first i with cum[i] >= roll.
 * Returns -1 where the reference would fall off the loop and return None. */
static int pick_allele_index(const oracle_snp* snp, double roll)
{
    for (int i = 0; i < snp->n_alleles; ++i)
        if (snp->cum[i] >= roll) return i;
    return -1;
}

int oracle_pick_allele_index(const oracle_snp* snp, double roll) { return pick_allele_index(snp, roll); }

/* is_haploid, common/snp.py:102-109 */
static int is_haploid(const char* chromo, int is_male)
{
    return (strcmp(chromo, "X") == 0 && is_male) || strcmp(chromo, "MT") == 0 || strcmp(chromo, "Y") == 0;
}

/* `snp.id not in sample.deleterious_snps`, pop_factory.py:485.  A sample's
 * carried ids are del_ids[del_off[i] .. del_off[i+1]). */
static int carries(const uint64_t* del_off, const int64_t* del_ids, uint64_t i, int64_t id)
{
    for (uint64_t k = del_off[i]; k < del_off[i + 1]; ++k)
        if (del_ids[k] == id) return 1;
    return 0;
}

/*
 * Rows [0, n_rows) of `snps` for S samples.  RNG row index of row j is
 * rng_row0 + j (the reference worker draws rows sequentially from one stream,
 * pop_factory.py:474-477; in MT mode the stream is advanced by 2*S doubles
 * per row, so rng_row0 rows are skipped first).
 *
 * sex[i] in {1,2} (SampleInfo.sex, pop_factory.py:70-71); is_control[i];
 * text receives the VCF body lines back to back; row_off[n_rows+1] their
 * start offsets; packed (optional) [n_rows][ceil(S/16)] u32, sample i at
 * bits 2*(i%16): bit0 = first allele != 0, bit1 = second allele != 0
 * (0 for haploid second allele and for '.').
 *
 * Returns 0, or -1 text_cap too small, -2 pick_allele_index fell through
 * (reference would raise TypeError at pop_factory.py:489/494), -3 alloc.
 */
int oracle_generate(const oracle_snp* snps, uint64_t n_rows, uint64_t rng_row0,
                    const uint8_t* sex, const uint8_t* is_control,
                    const uint64_t* del_off, const int64_t* del_ids, uint64_t S,
                    int rng_mode, uint64_t seed, const uint32_t* mt_key, int mt_pos,
                    char* text, uint64_t text_cap, uint64_t* row_off, uint32_t* packed)
{
    double* randoms = (double*)malloc(sizeof(double) * (2 * S + 4));
    if (!randoms) return -3;
    mt_state mt;
    if (rng_mode == ORACLE_RNG_MT19937) {
        if (mt_key) { memcpy(mt.key, mt_key, sizeof mt.key); mt.pos = mt_pos; }
        else mt_seed(&mt, (uint32_t)seed);
        for (uint64_t r = 0; r < rng_row0; ++r)
            for (uint64_t i = 0; i < 2 * S; ++i) (void)mt_double(&mt);
    }
    const uint64_t W = (S + 15) / 16;
    uint64_t pos = 0;
    int rc = 0;
    for (uint64_t j = 0; j < n_rows && rc == 0; ++j) {
        const oracle_snp* snp = &snps[j];
        /* pop_factory.py:477  randoms = numpy.random.rand(num_samples * 2) */
        if (rng_mode == ORACLE_RNG_MT19937)
            for (uint64_t i = 0; i < 2 * S; ++i) randoms[i] = mt_double(&mt);
        else
            philox_row_uniforms(seed, rng_row0 + j, 2 * S, randoms);

        row_off[j] = pos;
        /* worst case per row: prefix < 200 bytes, cell <= 4 bytes */
        if (pos + 256 + 4 * S + 2 > text_cap) { rc = -1; break; }
        /* pop_factory.py:503-507 */
        pos += (uint64_t)sprintf(text + pos, "%s\t%lld\trs%lld\t%s\t%s\t40\tPASS\t.\tGT\t", snp->chrom,
                                 (long long)snp->position, (long long)snp->id, snp->ref, snp->alt);
        uint32_t* prow = packed ? packed + j * W : NULL;
        if (prow) memset(prow, 0, W * sizeof(uint32_t));
        for (uint64_t i = 0; i < S; ++i) {
            int is_male = sex[i] == 1;              /* pop_factory.py:479 */
            int a = 0, b = 0, n_out;                /* n_out: 0 '.', 1 haploid, 2 diploid */
            if (!is_male && strcmp(snp->chrom, "Y") == 0) {
                n_out = 0;                          /* pop_factory.py:481-484 */
            } else if (is_control[i] || !carries(del_off, del_ids, i, snp->id)) {
                a = pick_allele_index(snp, randoms[2 * i]);          /* :486-487 */
                if (a < 0) { rc = -2; break; }
                if (is_haploid(snp->chrom, is_male)) {
                    n_out = 1;                                       /* :488-490 */
                } else {
                    b = pick_allele_index(snp, randoms[2 * i + 1]);  /* :492-493 */
                    if (b < 0) { rc = -2; break; }
                    n_out = 2;                                       /* :494 */
                }
            } else {
                a = 1;                              /* pop_factory.py:495-499 */
                if (is_haploid(snp->chrom, is_male)) n_out = 1;
                else { b = 1; n_out = 2; }
            }
            if (i) text[pos++] = '\t';              /* "\t".join(...)  :508 */
            if (n_out == 0) text[pos++] = '.';
            else if (n_out == 1) text[pos++] = (char)('0' + a);      /* str(selected_nt), a < ORACLE_MAX_ALLELES */
            else { text[pos++] = (char)('0' + a); text[pos++] = '/'; text[pos++] = (char)('0' + b); } /* "%i/%i" */
            if (prow) {
                uint32_t bits = (n_out >= 1 && a != 0 ? 1u : 0u) | (n_out == 2 && b != 0 ? 2u : 0u);
                prow[i / 16] |= bits << (2 * (i % 16));
            }
        }
        text[pos++] = '\n';
    }
    row_off[n_rows] = pos;
    free(randoms);
    return rc;
}

/* Same rows, several host threads: contiguous row ranges, one per thread
 * (Philox mode only: the counter-based stream makes ranges independent).
 * Used only by bench.py's cpu_baseline leg.  Text is discarded per thread
 * (each thread formats into its own scratch buffer) -- the timing covers
 * draw + cell logic + line formatting, as in the reference worker. */
#include <pthread.h>
typedef struct {
    const oracle_snp* snps; uint64_t n_rows, rng_row0;
    const uint8_t *sex, *is_control; const uint64_t* del_off; const int64_t* del_ids;
    uint64_t S, seed; uint64_t bytes; int rc;
} mt_job;

static void* mt_worker(void* p)
{
    mt_job* j = (mt_job*)p;
    const uint64_t chunk = 64;
    uint64_t cap = chunk * (4 * j->S + 300) + 1024;
    char* text = (char*)malloc(cap);
    uint64_t* off = (uint64_t*)malloc(sizeof(uint64_t) * (chunk + 1));
    j->bytes = 0; j->rc = 0;
    if (!text || !off) { j->rc = -3; free(text); free(off); return NULL; }
    for (uint64_t r = 0; r < j->n_rows && j->rc == 0; r += chunk) {
        uint64_t n = j->n_rows - r < chunk ? j->n_rows - r : chunk;
        j->rc = oracle_generate(j->snps + r, n, j->rng_row0 + r, j->sex, j->is_control, j->del_off, j->del_ids,
                                j->S, ORACLE_RNG_PHILOX, j->seed, NULL, 0, text, cap, off, NULL);
        j->bytes += off[n];
    }
    free(text); free(off);
    return NULL;
}

int oracle_generate_threads(const oracle_snp* snps, uint64_t n_rows, uint64_t rng_row0,
                            const uint8_t* sex, const uint8_t* is_control,
                            const uint64_t* del_off, const int64_t* del_ids, uint64_t S,
                            uint64_t seed, int n_threads, uint64_t* total_bytes)
{
    if (n_threads < 1) n_threads = 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
    mt_job* jobs = (mt_job*)malloc(sizeof(mt_job) * (size_t)n_threads);
    if (!th || !jobs) { free(th); free(jobs); return -3; }
    uint64_t per = (n_rows + (uint64_t)n_threads - 1) / (uint64_t)n_threads;
    for (int t = 0; t < n_threads; ++t) {
        uint64_t b = per * (uint64_t)t, e = b + per;
        if (b > n_rows) b = n_rows;
        if (e > n_rows) e = n_rows;
        jobs[t] = (mt_job){snps + b, e - b, rng_row0 + b, sex, is_control, del_off, del_ids, S, seed, 0, 0};
        pthread_create(&th[t], NULL, mt_worker, &jobs[t]);
    }
    int rc = 0; uint64_t tot = 0;
    for (int t = 0; t < n_threads; ++t) {
        pthread_join(th[t], NULL);
        if (jobs[t].rc) rc = jobs[t].rc;
        tot += jobs[t].bytes;
    }
    if (total_bytes) *total_bytes = tot;
    free(th); free(jobs);
    return rc;
}

/* Verify a buffer produced elsewhere (the HIP path) against this oracle, row range by row range on
 * n_threads host threads (Philox mode).  text/row_off/packed are the buffers under test, laid out like
 * oracle_generate's outputs.  *first_bad = index of the first differing row, or UINT64_MAX. */
typedef struct {
    const oracle_snp* snps; uint64_t row_lo, row_hi, rng_row0;
    const uint8_t *sex, *is_control; const uint64_t* del_off; const int64_t* del_ids;
    uint64_t S, seed; const char* text; const uint64_t* row_off; const uint32_t* packed;
    uint64_t bad; int rc;
} vf_job;

static void* vf_worker(void* p)
{
    vf_job* j = (vf_job*)p;
    const uint64_t chunk = 64, W = (j->S + 15) / 16;
    uint64_t cap = chunk * (4 * j->S + 300) + 1024;
    char* text = (char*)malloc(cap);
    uint64_t* off = (uint64_t*)malloc(sizeof(uint64_t) * (chunk + 1));
    uint32_t* pk = j->packed ? (uint32_t*)malloc(sizeof(uint32_t) * chunk * (W ? W : 1)) : NULL;
    j->bad = UINT64_MAX; j->rc = 0;
    if (!text || !off || (j->packed && !pk)) { j->rc = -3; free(text); free(off); free(pk); return NULL; }
    for (uint64_t r = j->row_lo; r < j->row_hi && j->rc == 0 && j->bad == UINT64_MAX; r += chunk) {
        uint64_t n = j->row_hi - r < chunk ? j->row_hi - r : chunk;
        j->rc = oracle_generate(j->snps + r, n, j->rng_row0 + r, j->sex, j->is_control, j->del_off, j->del_ids,
                                j->S, ORACLE_RNG_PHILOX, j->seed, NULL, 0, text, cap, off, pk);
        if (j->rc) break;
        for (uint64_t k = 0; k < n; ++k) {
            uint64_t len = off[k + 1] - off[k];
            if (j->row_off[r + k + 1] - j->row_off[r + k] != len ||
                memcmp(j->text + j->row_off[r + k], text + off[k], len) != 0 ||
                (pk && memcmp(j->packed + (r + k) * W, pk + k * W, W * sizeof(uint32_t)) != 0)) {
                j->bad = r + k;
                break;
            }
        }
    }
    free(text); free(off); free(pk);
    return NULL;
}

int oracle_verify_threads(const oracle_snp* snps, uint64_t n_rows, uint64_t rng_row0,
                          const uint8_t* sex, const uint8_t* is_control,
                          const uint64_t* del_off, const int64_t* del_ids, uint64_t S,
                          uint64_t seed, int n_threads, const char* text, const uint64_t* row_off,
                          const uint32_t* packed, uint64_t* first_bad)
{
    if (n_threads < 1) n_threads = 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
    vf_job* jobs = (vf_job*)malloc(sizeof(vf_job) * (size_t)n_threads);
    if (!th || !jobs) { free(th); free(jobs); return -3; }
    uint64_t per = (n_rows + (uint64_t)n_threads - 1) / (uint64_t)n_threads;
    for (int t = 0; t < n_threads; ++t) {
        uint64_t b = per * (uint64_t)t, e = b + per;
        if (b > n_rows) b = n_rows;
        if (e > n_rows) e = n_rows;
        jobs[t] = (vf_job){snps, b, e, rng_row0, sex, is_control, del_off, del_ids, S, seed, text, row_off, packed, UINT64_MAX, 0};
        pthread_create(&th[t], NULL, vf_worker, &jobs[t]);
    }
    int rc = 0; uint64_t bad = UINT64_MAX;
    for (int t = 0; t < n_threads; ++t) {
        pthread_join(th[t], NULL);
        if (jobs[t].rc) rc = jobs[t].rc;
        if (jobs[t].bad < bad) bad = jobs[t].bad;
    }
    if (first_bad) *first_bad = bad;
    free(th); free(jobs);
    return rc;
}
