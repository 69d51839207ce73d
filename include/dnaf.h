/*
 * dnaf.h -- C ABI of the MI355X row generator (libdnaf_hip.so).
 *
 * Drop-in boundary for dna-factory's per-SNP row-generation hot path.  The
 * reference has no FFI layer: the seam is two Python methods,
 *   PopulationFactory.queue_vcf_snps(fam_data, work_q, result_p)  pop_factory.py:471-513  (worker)
 *   PopulationFactory.write_vcf_snps(fam_data, snps, file)        pop_factory.py:417-469  (dispatcher)
 * and this header is what a ctypes binding for that seam binds (see
 * INTEGRATION.md for the reference-side stub).  Each entry point names the
 * reference lines whose work it takes over.
 *
 * Conventions: every function returns DNAF_OK (0) or a negative DNAF_ERR_*;
 * the message is available from dnaf_last_error().  No C++ exception crosses
 * the ABI and nothing aborts.  The caller owns every host buffer it passes;
 * no pointer is retained after a call returns.  The library owns the device
 * buffers inside a context.  A context is bound to one GPU and is not
 * thread-safe; different contexts may be used from different threads.
 */
#ifndef DNAF_H
#define DNAF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DNAF_ABI_VERSION 2 /* 2: the device copy of the packed matrix has a row pitch (dnaf_packed_pitch_words) */

#define DNAF_OK 0
#define DNAF_ERR_BAD_ARG (-1)
#define DNAF_ERR_HIP (-2)
#define DNAF_ERR_OOM (-3)
#define DNAF_ERR_UNSUPPORTED (-4)
#define DNAF_ERR_STATE (-5)
#define DNAF_ERR_THRESHOLD (-6) /* a roll could exceed the last cumulative frequency: the reference
                                   would raise TypeError at pop_factory.py:489/494 */

/* per-sample flag bits (SampleInfo.is_male / .is_control, pop_factory.py:47-71) */
#define DNAF_SAMPLE_MALE 1u
#define DNAF_SAMPLE_CONTROL 2u

/* chromosome class of a row: everything is_haploid() and the female-on-Y rule
 * look at (common/snp.py:102-109, pop_factory.py:481-484) */
#define DNAF_CLASS_AUTOSOME 0u /* diploid for everyone                      */
#define DNAF_CLASS_X 1u        /* haploid for males, diploid for females    */
#define DNAF_CLASS_Y 2u        /* haploid for males, '.' for females        */
#define DNAF_CLASS_MT 3u       /* haploid for everyone                      */

#define DNAF_MAX_ALLELES 4

/* RNG that stands in for numpy.random.rand(2*S) of pop_factory.py:477 */
#define DNAF_RNG_PHILOX 0u  /* Philox4x32-10, one 32-bit word per draw; rocRAND sequence
                               (seed, subsequence = RNG row, offset = 2*sample + allele)  */
#define DNAF_RNG_MT19937 1u /* numpy legacy stream (53-bit doubles from MT19937): the unmodified
                               reference after numpy.random.seed(seed); ranges must start on a
                               multiple of 156 / gcd(S, 156) rows                          */

/* what dnaf_generate materialises */
#define DNAF_OUT_TEXT 1u   /* VCF body lines, pop_factory.py:503-508                       */
#define DNAF_OUT_PACKED 2u /* 2-bit genotype matrix [n_rows][ceil(S/16)] u32               */

typedef struct dnaf_ctx dnaf_ctx;

/* SNP table in structure-of-arrays form: the fields of SNPTuples
 * (pop_factory.py:74-133) that the worker reads, for n_rows rows in output order. */
typedef struct dnaf_snp_table {
    uint64_t n_rows;
    const int64_t* snp_id;        /* [n_rows] SNPTuples.id (matched against the forced set)            */
    const uint8_t* chrom_class;   /* [n_rows] DNAF_CLASS_*                                             */
    const uint8_t* n_alleles;     /* [n_rows] len(tuples), 1..DNAF_MAX_ALLELES                         */
    const double* cum_freq;       /* [n_rows][DNAF_MAX_ALLELES] tuples[k][1]; unused entries ignored   */
    const uint8_t* prefix_bytes;  /* rows' fixed columns "chrom\tpos\trsID\tREF\tALT\t40\tPASS\t.\tGT\t" */
    const uint64_t* prefix_off;   /* [n_rows+1] offsets into prefix_bytes                              */
} dnaf_snp_table;

int dnaf_abi_version(void);
int dnaf_device_count(int* count);

int dnaf_create(int device, dnaf_ctx** out);
int dnaf_destroy(dnaf_ctx* ctx);
/* never NULL; valid until the next call on ctx (ctx == NULL: message of the last failed dnaf_create) */
const char* dnaf_last_error(const dnaf_ctx* ctx);

/* Sample table = the fam_data argument of queue_vcf_snps (pop_factory.py:471,
 * :478-485): flags[i] = DNAF_SAMPLE_* bits of sample i, in fam order.
 * Invalidates a previously set SNP table and forced set. */
int dnaf_set_samples(dnaf_ctx* ctx, const uint8_t* flags, uint64_t n_samples);

/* Forced ("deleterious") genotypes = SampleInfo.deleterious_snps of the case
 * samples (pop_factory.py:485, :495-499), transposed by SNP: for slot k,
 * samples carrier_idx[carrier_off[k] .. carrier_off[k+1]) carry SNP snp_ids[k].
 * Control samples must not be listed (they are never forced, :485).
 * n_slots == 0 clears the set.  Call after dnaf_set_samples and before
 * dnaf_set_snps. */
int dnaf_set_forced(dnaf_ctx* ctx, const int64_t* snp_ids, uint64_t n_slots,
                    const uint64_t* carrier_off, const uint32_t* carrier_idx);

/* RNG mode and seed.  rng_row0 = RNG row index of table row 0 (the global SNP
 * row index when the table is one shard of a larger job), so any shard of a
 * job reproduces the bytes of the whole job.  May be called at any time. */
int dnaf_set_rng(dnaf_ctx* ctx, uint32_t mode, uint64_t seed, uint64_t rng_row0);

/* MT19937 mode with an explicit generator state instead of a seed: key[624] and pos are
 * numpy.random.get_state()[1] and [2] at the moment the reference's worker would start drawing
 * (pop_factory.py:432-434, :477), so that a run whose host side replays the reference's earlier
 * numpy calls continues the very same stream. */
int dnaf_set_mt_state(dnaf_ctx* ctx, const uint32_t* key, uint32_t pos, uint64_t rng_row0);

/* Only the RNG row of table row 0 changes; mode, seed / generator state and whatever the library has derived from
 * them stay (in MT19937 mode: the cached block states, which a later window of the same stream is reached from by
 * one jump per wave).  For a context that is handed the tiles of a larger job one table at a time -- the way the
 * reference deals rows to its workers, stripe_list common/snp.py:82-88, pop_factory.py:426: here tile k of the job
 * goes to device k mod G. */
int dnaf_set_rng_row0(dnaf_ctx* ctx, uint64_t rng_row0);

/* Table rows k*n .. k*n + n-1 share RNG row rng_row0 + k (n = 1, the default: every row its own).  This is what the
 * reference's `-n n` does to the stream: its n forked workers all inherit the parent's numpy state and never reseed, so
 * worker w draws for rows w+1, w+1+n, ... the very numbers worker 0 draws for rows 1, 1+n, ... (stripe_list
 * common/snp.py:82-88, fork pop_factory.py:426-434).  Kept as a compatibility mode (--compat-workers) so that
 * the reference's DEFAULT invocation (-n 2) can be reproduced seed for seed.  A range passed to dnaf_generate must start on
 * a multiple of n. */
int dnaf_set_rng_divisor(dnaf_ctx* ctx, uint32_t n);

/* Upload the SNP table (the work_q argument, pop_factory.py:474) and lay out
 * the output: per-row integer thresholds, forced slot, text offsets. */
int dnaf_set_snps(dnaf_ctx* ctx, const dnaf_snp_table* table);

/* Bytes of VCF text that rows [row_begin, row_begin+n_rows) produce. */
int dnaf_text_bytes(const dnaf_ctx* ctx, uint64_t row_begin, uint64_t n_rows, uint64_t* bytes);
/* Start offsets of those rows relative to the first one: off[n_rows+1]. */
int dnaf_row_offsets(const dnaf_ctx* ctx, uint64_t row_begin, uint64_t n_rows, uint64_t* off);

/* Generate rows [row_begin, row_begin+n_rows) of the table into the
 * context's device buffers (what = DNAF_OUT_* bits).  This is the worker
 * loop pop_factory.py:474-509 for those rows.  Asynchronous on the
 * context's stream; grows the output buffers if needed (a growth
 * synchronises the device). */
int dnaf_generate(dnaf_ctx* ctx, uint64_t row_begin, uint64_t n_rows, uint32_t what);
/* Size the output buffers ahead of time so that no later dnaf_generate has to grow them: text_bytes of
 * VCF text and packed_words = rows * ceil(S/16) words of the matrix (dense count; the library adds the
 * padding of its device pitch itself). */
int dnaf_reserve_output(dnaf_ctx* ctx, uint64_t text_bytes, uint64_t packed_words);
int dnaf_sync(dnaf_ctx* ctx);
/* Name of the row kernel the last dnaf_generate launched ("rowrun_kernel", "rowmix_kernel", "rowgen_kernel",
 * "mt_rowgen_kernel"; "" before the first call): which form of the worker loop (pop_factory.py:471-513) a measurement
 * is a measurement of. */
int dnaf_last_kernel(dnaf_ctx* ctx, char* name, uint64_t cap);

/* Device time (HIP events on the context's stream) of the last
 * dnaf_generate, in milliseconds.  Synchronises. */
int dnaf_last_generate_ms(dnaf_ctx* ctx, float* ms);

/* Per-kernel device timing over a region: dnaf_timing_begin() resets the
 * recorder; every dnaf_generate() after it records a HIP event pair around
 * its kernel on the context's stream (at most 4096 launches are recorded);
 * dnaf_timing_end() synchronises and returns the summed kernel time, the
 * number of launches recorded and the number of launches made. */
int dnaf_timing_begin(dnaf_ctx* ctx);
int dnaf_timing_end(dnaf_ctx* ctx, double* total_ms, uint64_t* n_recorded, uint64_t* n_launches);

/* Copy the last generated range to host memory (synchronises).
 * text: exactly dnaf_text_bytes() bytes; packed: n_rows*ceil(S/16) u32. */
int dnaf_read_text(dnaf_ctx* ctx, void* dst, uint64_t dst_cap, uint64_t* n_bytes);
int dnaf_read_packed(dnaf_ctx* ctx, uint32_t* dst, uint64_t dst_cap_words, uint64_t* n_words);
/* Per-row alternative-allele counts of the last generated packed matrix, computed on the GPU:
 * counts[2*j] = alternative alleles among the control samples of row j, counts[2*j+1] among the cases
 * (the case/control frequency check of BASELINE config 5).  Synchronises. */
int dnaf_allele_counts(dnaf_ctx* ctx, uint32_t* counts, uint64_t cap_rows);
/* Bytes [byte_begin, byte_begin + n_bytes) of the last generated text (consumers that stream a tile out in pieces). */
int dnaf_read_text_range(dnaf_ctx* ctx, uint64_t byte_begin, uint64_t n_bytes, void* dst);

/* PLINK 1 binary genotypes of the last generated packed matrix, made on the GPU: n_rows rows of
 * ceil(S/4) bytes in .bed order (SNP-major; sample i at bits 2*(i%4) of byte i/4; with A1 = ALT and
 * A2 = REF in the .bim: 3 = 0/0, 2 = heterozygous, 0 = 1/1, 1 = missing).  Haploid cells (males on X / Y,
 * MT) are stored as homozygous, females on Y as missing -- what PLINK makes of the VCF the reference
 * writes (plink_analysis.py feeds population.vcf.gz to plink).  The 3 magic bytes are the caller's. */
int dnaf_export_bed(dnaf_ctx* ctx, uint64_t* n_bytes);
int dnaf_read_bed(dnaf_ctx* ctx, void* dst, uint64_t dst_cap, uint64_t* n_bytes);

/* Device pointers of the last generated range, for consumers that stay on
 * the GPU (valid until the next dnaf_generate / dnaf_destroy).  In device memory row j of the
 * matrix starts dnaf_packed_pitch_words() words after row j-1 and holds ceil(S/16) words;
 * packed_words = rows * pitch.  (dnaf_read_packed returns the dense matrix.) */
int dnaf_device_buffers(dnaf_ctx* ctx, void** d_text, uint64_t* text_bytes, void** d_packed, uint64_t* packed_words);
/* Row pitch of the device copy of the packed matrix, in 32-bit words: ceil(S/16) rounded up to whole
 * 128-byte lines once a row is at least 512 bytes long (so that every line of the matrix is stored by
 * exactly one wave), ceil(S/16) below that.  Known after dnaf_set_samples. */
int dnaf_packed_pitch_words(const dnaf_ctx* ctx, uint64_t* pitch_words);

/* On-GPU BGZF encoder -- the sink right after the path (the reference pushes every line through
 * Bio.bgzf.BgzfWriter, pop_factory.py:403, :449).  Compresses the text of the last dnaf_generate
 * into a sequence of BGZF members (gzip members with the BC extra field, <= 64 KiB each, no EOF
 * block) in device memory; any gzip reader returns the input bytes.  When the range was generated
 * with DNAF_OUT_TEXT | DNAF_OUT_PACKED, members that lie in biallelic autosome rows are encoded from
 * the 2-bit matrix (LZ77 over cells, about 30 % smaller output); the decompressed bytes are the same.
 * flags: DNAF_BGZF_REUSE_TABLES keeps the Huffman tables of the previous call instead of sampling
 * the new rows.  Synchronises. */
#define DNAF_BGZF_REUSE_TABLES 1u
#define DNAF_BGZF_TEXT_DOMAIN 2u /* encode every member from the text (no genotype-domain members) */
int dnaf_bgzf_compress(dnaf_ctx* ctx, uint32_t flags, uint64_t* n_bytes);
int dnaf_read_bgzf(dnaf_ctx* ctx, void* dst, uint64_t dst_cap, uint64_t* n_bytes);
/* The same data through the context's pinned staging buffers (DMA at PCIe rate instead of a pageable
 * copy): *host points at n_bytes bytes owned by the library.  Four staging buffers are used in turn, so the
 * pointer stays valid through the next THREE dnaf_fetch_* calls and is reused by the fourth (writer threads
 * can drain ranges k-2 .. k while range k+1 is generated and fetched).  Synchronises. */
int dnaf_fetch_bgzf(dnaf_ctx* ctx, const void** host, uint64_t* n_bytes);
int dnaf_fetch_text(dnaf_ctx* ctx, const void** host, uint64_t* n_bytes);
/* Summed device time of the encoder's kernels in the last dnaf_bgzf_compress (HIP events). */
int dnaf_last_bgzf_ms(dnaf_ctx* ctx, float* ms);
/* Host-only helper: the encoder's code tables for a token histogram (256 literal counts, 256 counts
 * of match lengths 3..258, end-of-block count); out receives an opaque table blob. */
/* Host helper: the lines of snps.json.gz, one str(SNPTuples) per row (pop_factory.py:118-124, :258-262).
 * Numbers are passed as indices into a table of formatted values (repr(float) per distinct value). */
int dnaf_format_snps_json(uint64_t n, const int64_t* id, const char* chrom, uint32_t chrom_w, const int64_t* pos,
                          const char* ref, uint32_t ref_w, const char* alt, uint32_t alt_w, const uint8_t* n_alleles,
                          const int32_t* cum_idx, const char* num_blob, const uint64_t* num_off, uint8_t* out,
                          uint64_t out_cap, uint64_t* n_bytes);

/* Host helper: the inverse, for files written by dna-factory or by this library (SNPTuples.from_json,
 * pop_factory.py:126-133).  alleles: [max_rows][4] NUL-padded fields of allele_w bytes in "tuples" key order.
 * DNAF_ERR_UNSUPPORTED = a line this fast path does not understand (use a JSON parser instead). */
int dnaf_parse_snps_json(const char* text, uint64_t n_bytes, uint64_t max_rows, int64_t* id, char* chrom, uint32_t chrom_w,
                         int64_t* pos, char* alleles, uint32_t allele_w, double* cum, uint8_t* n_alleles, uint64_t* n_rows);

/* Host-only helpers of the MT19937 mode (tests): the window of 624 untempered words J >= 1 words after
 * base[624]; the characteristic polynomial of the generator (312 coefficient words, degree). */
int dnaf_mt_jump_host(const uint32_t* base, uint64_t J, uint32_t* window);
/* Host helper of the seed-compatible front end: numpy.random.choice(pop_size, size, replace=False) of numpy's legacy
 * RandomState, as the reference's deleterious picks call it (pop_factory.py:576).  key[624] / *pos are
 * numpy.random.get_state()[1] / [2] and are advanced exactly as numpy would advance them; out_idx[size] receives
 * permutation(pop_size)[:size].  Replays the draws of numpy's shuffle without shuffling (the wanted entries are traced back
 * through the swaps).  DNAF_ERR_UNSUPPORTED: size > pop_size or pop_size >= 2^32 (let numpy handle / raise). */
int dnaf_numpy_legacy_choice(uint32_t* key, uint32_t* pos, uint64_t pop_size, uint32_t size, uint64_t* out_idx);
int dnaf_mt_characteristic_polynomial(uint64_t* words312, int* degree);
int dnaf_bgzf_tables_from_histogram(const uint64_t* lit_hist, const uint64_t* match_hist, uint64_t n_eob, void* out,
                                    uint64_t out_cap, uint64_t* size);
/* Host-only helper: the genotype-domain encoder's tables for hist[8][320] -- per table class 286
 * literal/length symbol counts followed by 30 distance symbol counts (4 unused); out receives an
 * opaque blob (code tables and run-length coded block header per class, CRC-32 unit tables). */
int dnaf_gtz_tables_from_histogram(const uint32_t* hist, void* out, uint64_t out_cap, uint64_t* size);

/* Host helper: format the fixed columns of each row exactly like
 * pop_factory.py:503-507 ("%s\t%i\trs%s\t%s\t%s\t40\tPASS\t.\tGT\t").
 * chrom/ref/alt are fixed-width, NUL-padded byte fields (numpy 'S<w>' arrays).
 * Pass out == NULL to get only the offsets (off[n] = bytes needed). */
int dnaf_format_prefixes(uint64_t n, const char* chrom, uint32_t chrom_w, const int64_t* pos, const int64_t* id,
                         const char* ref, uint32_t ref_w, const char* alt, uint32_t alt_w, uint8_t* out,
                         uint64_t out_cap, uint64_t* off);

#ifdef __cplusplus
}
#endif
#endif /* DNAF_H */
