/*
 * include/hydia.h — C-ABI of libhydia.so: the MI355X-native drop-in for the HyDia (approach 5) path of
 * n7koirala/image_matching.  Plain pointers and sizes only; no torch / OpenFHE / C++ types cross this boundary.
 *
 * The reference has no FFI: its seam is the C++ virtual surface main.cpp programs against
 * (/root/reference/include/sender.h:19-43, include/receiver.h:17-43, include/enroller_diag.h:7-27), carrying OpenFHE
 * shared_ptr handles.  Each entry point below names the reference interface it replaces; INTEGRATION.md shows the
 * adapter a maintainer adds on the reference side.
 *
 * Data crossing the boundary:
 *   ciphertext  = uint64 residues, limb-major [poly][limb][N], EVALUATION form in bit-reversed order
 *                 (out[j] = a(psi^(2*bitrev(j)+1)) mod q_limb), limb j <-> modulus j of hydia_get_moduli();
 *                 what OpenFHE exposes as ct->GetElements()[p].GetElementAtIndex(j).GetValues() after
 *                 SetFormat(EVALUATION) when the moduli/roots agree (otherwise cross in COEFFICIENT form and
 *                 convert with hydia_ntt — INTEGRATION.md).
 *   eval key    = [digit][2][limb over Q then P][N] residues, poly 0 = b, poly 1 = a (hybrid key switching, dnum digits)
 *   slots       = IEEE doubles
 * All functions return 0 on success and a negative hydia_status otherwise; hydia_last_error() has the message
 * (the reference prints to cerr and carries on — src/sender/sender_diag.cpp:89-91 — callers that want that behaviour
 * ignore the code).  One host thread per context at a time; contexts are independent (one per GPU, or several per GPU) and
 * every entry point selects its context's GPU itself.  A context stays alive until its last hydia_ct handle is freed.
 */
#ifndef HYDIA_H
#define HYDIA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    HYDIA_OK = 0,
    HYDIA_ERR_ARG = -1,      /* bad argument / shape */
    HYDIA_ERR_STATE = -2,    /* missing key / database / wrong level */
    HYDIA_ERR_DEVICE = -3,   /* HIP failure (no GPU, out of memory, ...) */
    HYDIA_ERR_INTERNAL = -4
} hydia_status;

/* CKKS parameters — the knobs of /root/reference/src/main.cpp:169-173 plus VECTOR_DIM (include/config.h:30). */
typedef struct {
    uint32_t log_n;          /* 15  (HEStd_128_classic at this modulus size) */
    uint32_t mult_depth;     /* 11  = OpenFHEWrapper::computeRequiredDepth(5), src/openFHE_wrapper.cpp:37-40 */
    uint32_t scale_bits;     /* 45  SetScalingModSize */
    uint32_t first_mod_bits; /* 60  OpenFHE default first modulus */
    uint32_t dnum;           /* 3   OpenFHE default numLargeDigits (HYBRID) */
    uint32_t vector_dim;     /* 512 VECTOR_DIM */
} hydia_params;

typedef struct {
    uint32_t log_n, n, slots, n_q, n_p, dnum, alpha, vector_dim;
    double delta;            /* 2^scale_bits */
} hydia_info;

typedef struct hydia_ctx hydia_ctx;
typedef struct hydia_ct hydia_ct; /* a batch of >= 1 ciphertexts of identical shape, resident in HBM */

const char *hydia_last_error(void);
const char *hydia_version(void);

/* hydia_default_params: the context of ./ImageMatching <file> 5 (src/main.cpp:82, :169-173). */
void hydia_default_params(hydia_params *out);
/* Host-only parameter derivation (no GPU needed): moduli/roots are length n_q + n_p, Q limbs first. */
int hydia_params_describe(const hydia_params *p, hydia_info *info, uint64_t *moduli, uint64_t *roots);
/* OpenFHEWrapper::computeRequiredDepth, src/openFHE_wrapper.cpp:6-44 */
size_t hydia_compute_required_depth(size_t approach);

/* replaces GenCryptoContext + Enable(...) (src/main.cpp:169-179) for the sender/receiver on GPU `device`.  No HIP device, or a
 * device index beyond the ones visible: HYDIA_ERR_DEVICE (a negative index: HYDIA_ERR_ARG) — there is no CPU fallback. */
int hydia_ctx_create(const hydia_params *p, int device, hydia_ctx **out);
/* Same, on a caller-supplied prime chain — the adapter path of SURVEY 8f-3: an OpenFHE context's ciphertext primes
 * (cc->GetElementParams()->GetParams()[j]->GetModulus(), q_0 first) followed by its special primes
 * (GetParamsP()), and optionally the 2N-th roots OpenFHE uses (GetRootOfUnity()) so evaluation-form data can cross the
 * boundary unconverted.  n_q = mult_depth + 1.  Every modulus must be a distinct prime < 2^60 that is 1 mod 2N; limbs
 * of at most 47 bits take the FP64 NTT path; limbs below 2^48 are stored packed in the database (46-bit residues in the
 * group-sequential layout when every limb but q_0 is below 2^46, 48-bit otherwise: hydia_db_residue_bits).
 * The fused key-switching pipeline (column-fused conversions) serves up to four special primes of any width — OpenFHE's choice for the
 * reference's parameter set — or five below 2^48; other counts run the same arithmetic through the unfused kernels (bit-identical, slower). */
int hydia_ctx_create_custom(const hydia_params *p, const uint64_t *moduli, const uint64_t *roots /* may be NULL */,
                            uint32_t n_q, uint32_t n_p, int device, hydia_ctx **out);
void hydia_ctx_destroy(hydia_ctx *ctx);
int hydia_get_info(const hydia_ctx *ctx, hydia_info *out);
int hydia_get_moduli(const hydia_ctx *ctx, uint64_t *moduli, uint64_t *roots);
int hydia_sync(hydia_ctx *ctx);
int hydia_memory_stats(hydia_ctx *ctx, uint64_t *pool_live, uint64_t *pool_cached, uint64_t *pool_peak);

/* ---- randomness.  Every seed below is a 32-byte ChaCha20 key; samples are addressed by (seed, nonce), so a (seed, nonce) pair
 * must NEVER be used for two different plaintexts (the two ciphertexts would differ by exactly the plaintext difference).
 * The reference draws OpenFHE's PRNG seed from the OS; callers that do not need reproducibility do the same with
 * hydia_random_seed (getrandom(2)).  Nonces are 40-bit: encryption calls refuse nonce (+ count) >= 2^40. */
int hydia_random_seed(uint8_t out[32]);

/* ---- keys: cc->KeyGen / EvalMultKeyGen / EvalRotateKeyGen (src/main.cpp:184-206) ---- */
/* generate sk, pk, relin key and rotation keys {1..dim-1} u {dim, 2dim, .., slots/2} on the GPU from a 32-byte seed */
int hydia_keygen(hydia_ctx *ctx, const uint8_t seed[32]);
/* or import keys produced elsewhere (the reference's serial/{multkey,rotkey}.bin contents after unmarshalling):
 * rot = 0 is the relinearisation key, rot >= 1 the key of EvalRotate(., rot); data [dnum][2][n_q+n_p][N] */
int hydia_import_eval_key(hydia_ctx *ctx, int rot, const uint64_t *data);
int hydia_export_eval_key(hydia_ctx *ctx, int rot, uint64_t *data);
int hydia_import_public_key(hydia_ctx *ctx, const uint64_t *data /* [2][n_q][N] (b, a) */);
int hydia_import_secret_key(hydia_ctx *ctx, const uint64_t *data /* [n_q+n_p][N], evaluation form */);
int hydia_export_public_key(hydia_ctx *ctx, uint64_t *data);
int hydia_export_secret_key(hydia_ctx *ctx, uint64_t *data);
int hydia_has_eval_key(hydia_ctx *ctx, int rot);
/* profiling filler: relin + rotation keys {1..dim-1} u {dim..slots/2 powers of two} of uniformly random residues
 * (kernel cost is data independent; results decrypt to noise) */
int hydia_fill_eval_keys_random(hydia_ctx *ctx, uint64_t seed);

/* ---- ciphertext handles (Ciphertext<DCRTPoly>) ---- */
int hydia_ct_import(hydia_ctx *ctx, const uint64_t *data, uint32_t count, uint32_t n_polys, uint32_t n_limbs,
                    double scale, hydia_ct **out);
int hydia_ct_export(hydia_ctx *ctx, const hydia_ct *ct, uint64_t *data);
int hydia_ct_shape(const hydia_ct *ct, uint32_t *count, uint32_t *n_polys, uint32_t *n_limbs, double *scale);
/* raw HBM address of the batch (for RCCL gathers through torch.distributed; layout [count][poly][limb][N]) */
int hydia_ct_device_ptr(const hydia_ct *ct, void **ptr, size_t *bytes);
/* device-to-device copy of the whole batch into caller-owned HBM (e.g. a torch tensor used as RCCL send buffer) */
int hydia_ct_copy_to_device(hydia_ctx *ctx, const hydia_ct *ct, void *dev_dst);
int hydia_ct_from_device(hydia_ctx *ctx, const void *dev_ptr, uint32_t count, uint32_t n_polys, uint32_t n_limbs,
                         double scale, hydia_ct **out); /* copies */
/* a handle over ciphertexts that STAY in the caller's device memory (no copy): the all-gathered rotations of a rotation-split loop A.
 * The memory must stay valid and unchanged while the handle, or work enqueued on it, is alive */
int hydia_ct_view_device(hydia_ctx *ctx, void *dev_ptr, uint32_t count, uint32_t n_polys, uint32_t n_limbs, double scale,
                         hydia_ct **out);
void hydia_ct_free(hydia_ct *ct);

/* ---- receiver: DiagonalReceiver / HersReceiver ---- */
/* Receiver::encryptQuery, src/receiver/receiver_diag.cpp:13-26: normalise, tile to all slots, encode, encrypt */
int hydia_encrypt_query(hydia_ctx *ctx, const double *query /* vector_dim */, const uint8_t seed[32], uint64_t nonce,
                        hydia_ct **out);
/* OpenFHEWrapper::encryptFromVector, src/openFHE_wrapper.cpp:74-77 (count vectors of `slots` doubles each) */
int hydia_encrypt(hydia_ctx *ctx, const double *slots, uint32_t count, const uint8_t seed[32], uint64_t nonce0,
                  hydia_ct **out);
/* OpenFHEWrapper::decryptToVector, src/openFHE_wrapper.cpp:81-85: out = count * slots doubles */
int hydia_decrypt(hydia_ctx *ctx, const hydia_ct *ct, double *out);
/* HersReceiver::decryptMembership, src/receiver/receiver_hers.cpp:26-35: slot 0 >= 1.0 */
int hydia_decrypt_membership(hydia_ctx *ctx, const hydia_ct *ct, int *result);
/* HersReceiver::decryptIndex, src/receiver/receiver_hers.cpp:37-54: every slot >= 1.0 -> j + i*slots.
 * *n_out receives the number of matches; at most cap are written. */
int hydia_decrypt_index(hydia_ctx *ctx, const hydia_ct *cts, size_t *out, size_t cap, size_t *n_out);

/* ---- enroller: DiagonalEnroller ---- */
/* number of DB ciphertexts for n vectors (concatenateRows, src/enroller/enroller_diag.cpp:120-122) */
size_t hydia_db_num_cts(const hydia_ctx *ctx, size_t n_vectors);
/* DiagonalEnroller::serializeDB, src/enroller/enroller_diag.cpp:12-53: normalises db IN PLACE (like the reference),
 * diagonalises, encodes and encrypts straight into the HBM-resident layout (no serial/db_diagonal files).
 * For a multi-GPU database each rank enrols its own contiguous range of 16384-vector blocks (multi-GPU). */
int hydia_db_enroll(hydia_ctx *ctx, double *db /* n x vector_dim row-major */, size_t n, const uint8_t seed[32]);
/* One shard of a database that is cut by 16384-vector row-blocks over several contexts (multi-GPU): `db` holds only
 * this shard's rows and first_block is the index of its first block in the whole database, so the shard encrypts with exactly
 * the nonces the unsharded enrolment uses for those blocks (bit-identical ciphertexts). */
int hydia_db_enroll_shard(hydia_ctx *ctx, double *db, size_t n, const uint8_t seed[32], size_t first_block);
/* ---- the split of the diagonalised mat-vec.  With rotation i = b + B g: B - 1 hoisted ("baby") rotations of the
 * query per QUERY, vector_dim / B relinearised partial sums per BLOCK of which all but the first are rotated by B g ("giant" steps,
 * ordinary key switches with the rotation keys B, 2B, .. that src/main.cpp:195-206 already generates).  The enroller rotates
 * diagonal i by -B (i div B) slots in the clear; ciphertext order and nonces do not change.
 *   B = vector_dim      "hoisted": the reference's own form (src/sender/sender_diag.cpp:22-26), no pre-rotation, no giant step
 *   B = 32 (dim 512)    "bsgs": the classic square-root split BASELINE.json's north_star names
 *   any power of two dividing vector_dim in between
 * Decrypted results agree within CKKS noise (1e-4 on scores) whatever B; ciphertexts are bit-identical between runs with the same B.
 * hydia_set_matvec mode: 0 auto (hydia_auto_babies: B grows with the blocks the enrolling context holds — at vector_dim 512: 64 up
 * to 3 blocks, 128 up to 12, 256 up to 40, hoisted above; measured, profiles/r04/matvec_sweep.txt), 1 hoisted, otherwise B itself; initial value from HYDIA_MATVEC=auto|hoisted|bsgs|<B>.
 * It takes effect at the NEXT enrolment; hydia_db_kind / hydia_db_babies tell what is resident (kind 0 none, 5 hoisted diagonals,
 * 6 pre-rotated diagonals, 4 HERS columns).  Ciphertexts imported one by one (hydia_db_alloc + hydia_db_import_ct: the reference
 * enroller's) are taken as hoisted unless hydia_db_set_babies says otherwise (a database of more than 8 blocks is then re-ordered in
 * HBM for the declared form, through a second buffer of its size — see hydia_db_group).  hydia_db_set_babies takes a DECLARED form:
 * vector_dim (hoisted) or a power of two >= 2 dividing it — 0, 1 and anything else are HYDIA_ERR_ARG; without a diagonal database
 * HYDIA_ERR_STATE; when the second buffer does not fit HYDIA_ERR_DEVICE, and the database, its layout and its form are untouched. */
int hydia_set_matvec(hydia_ctx *ctx, int mode);
int hydia_get_matvec(const hydia_ctx *ctx);
int hydia_db_kind(const hydia_ctx *ctx);
int hydia_db_babies(const hydia_ctx *ctx);
int hydia_db_set_babies(hydia_ctx *ctx, int babies);
int hydia_auto_babies(const hydia_ctx *ctx, size_t blocks); /* what an enrolment of `blocks` blocks on this context would pick */
/* hydia_db_enroll_shard with an explicit split (0 = the context's policy, 1 = hoisted, else B): a sharded enrolment passes ONE
 * decision to every shard */
int hydia_db_enroll_shard_ex(hydia_ctx *ctx, double *db, size_t n, const uint8_t seed[32], size_t first_block, int matvec);
/* or load ciphertexts produced elsewhere: t = block*vector_dim + diagonal, i.e. serial/db_diagonal/index<t>.bin
 * (src/enroller/enroller_diag.cpp:161; read back at src/sender/sender_diag.cpp:87-91) */
int hydia_db_alloc(hydia_ctx *ctx, size_t n_vectors);
int hydia_db_import_ct(hydia_ctx *ctx, size_t t, const uint64_t *data /* [2][n_q][N] */);
int hydia_db_export_ct(hydia_ctx *ctx, size_t t, uint64_t *data);
/* Persistence of the enrolled database (the reference keeps one serial/db_diagonal/index<t>.bin per ciphertext,
 * src/enroller/enroller_diag.cpp:158-166, and re-reads them every query; here the database stays in HBM and a file is only
 * what a server restart needs).  Own streaming format: a header (parameters, prime chain, packing) + the ciphertexts in order, each
 * as its packed residues (the ciphertext-major resident layout verbatim; a group-sequential database is converted on the way);
 * hydia_db_load refuses a file written for other parameters / primes / residue width. */
int hydia_db_save(hydia_ctx *ctx, const char *path);
int hydia_db_load(hydia_ctx *ctx, const char *path);
/* benchmark filler: n_vectors worth of uniformly random residues (the kernels' cost is data independent) */
int hydia_db_fill_random(hydia_ctx *ctx, size_t n_vectors, uint64_t seed);
int hydia_db_stats(const hydia_ctx *ctx, size_t *n_vectors, size_t *n_cts, size_t *bytes);
/* How the resident database lies in HBM: 0 = ciphertext after ciphertext; g > 0 = group-sequential, the layout a hoisted database of
 * more than 8 blocks takes — the bytes one loop-B workgroup reads (one 128-coefficient tile of one limb of g blocks) form one
 * sequential run, which HBM serves at 7.0 TB/s instead of 6.05.  Transparent to every entry point
 * (hydia_db_import_ct / hydia_db_export_ct address ciphertexts, hydia_db_save writes the ciphertext-major file format whatever the
 * resident layout); HYDIA_DB_CT_MAJOR=1 at context creation keeps every database ciphertext-major. */
int hydia_db_group(const hydia_ctx *ctx);
/* bits per stored residue of the 45/46-bit limbs of the resident database: 46 (group-sequential layout: 128 residues in a 736-byte
 * unit; round 4), 48 (6-byte residues: ciphertext-major layout, HYDIA_DB_48BIT, files) or 64 (HYDIA_DB_UNPACKED); 0 without a database.
 * Limb 0 (60 bit) always takes 8 bytes.  hydia_db_stats reports the bytes this makes resident. */
int hydia_db_residue_bits(const hydia_ctx *ctx);

/* ---- sender: DiagonalSender (src/sender/sender_diag.cpp) ---- */
/* loop A alone (:20-26): the vector_dim rotated queries, rot[0] = q */
int hydia_rotate_query(hydia_ctx *ctx, const hydia_ct *query, hydia_ct **out);
/* Sender::computeSimilarity (:12-33): one score ciphertext per 16384-vector block, level 1 */
int hydia_compute_similarity(hydia_ctx *ctx, const hydia_ct *query, hydia_ct **out);
/* Sender::indexScenario (:52-63) */
int hydia_index_scenario(hydia_ctx *ctx, const hydia_ct *query, hydia_ct **out);
/* Loop A split over the GPUs of a node (the `#pragma omp parallel for` over i of src/sender/sender_diag.cpp:23-26, cut into ranges):
 * rotations first .. first+count-1 of the query (rotation 0 = the query itself) written to dev_dst [count][2][n_q][N]; the ranges of
 * all ranks, all-gathered into one [vector_dim][2][n_q][N] buffer, are what hydia_rotate_query returns — and what the *_rotated
 * forms of computeSimilarity / indexScenario take instead of the query (a hydia_ct_view_device over the gathered buffer) */
int hydia_rotate_query_range(hydia_ctx *ctx, const hydia_ct *query, uint32_t first, uint32_t count, hydia_ct **out);
int hydia_rotate_query_range_into(hydia_ctx *ctx, const hydia_ct *query, uint32_t first, uint32_t count, void *dev_dst);
int hydia_compute_similarity_rotated(hydia_ctx *ctx, const hydia_ct *rotations, hydia_ct **out);
int hydia_index_scenario_rotated(hydia_ctx *ctx, const hydia_ct *rotations, hydia_ct **out);
/* Sender::membershipScenario (:35-50) */
int hydia_membership_scenario(hydia_ctx *ctx, const hydia_ct *query, hydia_ct **out);
/* OpenFHEWrapper::chebyshevCompare (src/openFHE_wrapper.cpp:143-185) on every ciphertext of the batch */
int hydia_chebyshev_compare(hydia_ctx *ctx, const hydia_ct *in, double delta, size_t sign_depth, hydia_ct **out);
/* multi-GPU membership tail: sum the batch into one ciphertext, then EvalSum over all slots (:46-47) */
int hydia_sum_and_evalsum(hydia_ctx *ctx, const hydia_ct *in, hydia_ct **out);

/* The two halves of that tail on their own — what a sharded membership query is composed of (sender_diag.cpp:46-47):
 * EvalAddManyInPlace over the batch -> ONE ciphertext; EvalSum(ct, batchSize) of one ciphertext. */
int hydia_add_many(hydia_ctx *ctx, const hydia_ct *in, hydia_ct **out);
int hydia_eval_sum(hydia_ctx *ctx, const hydia_ct *in, hydia_ct **out);
/* Cross-shard reduction of the partial sums: acc += src as plain 64-bit integers (src: same shape, compact, in HBM of
 * src_device; -1 = this context's GPU), and afterwards every value -> its canonical residue.  At most 16 residues below 2^60
 * fit 64 bits, so an RCCL all-reduce(SUM) on int64 over the handle's memory (hydia_ct_device_ptr) followed by
 * hydia_ct_mod_reduce is the multi-process form of the same step. */
int hydia_ct_add_raw(hydia_ctx *ctx, hydia_ct *acc, const void *dev_src, int src_device);
int hydia_ct_mod_reduce(hydia_ctx *ctx, hydia_ct *ct);

/* ---- sharded sender: one database over R contexts of ONE process (one per GPU of a node; shards may share a GPU) ----
 * Replaces the serial block loop of DiagonalSender::computeSimilarity (src/sender/sender_diag.cpp:28-30): shard r owns the
 * contiguous block range hydia_shard_blocks(G, R, r), every shard has the keys (same seed; one resident copy per GPU) and
 * runs loop A + its own mat-vec + comparator on its own host thread.  Queries enter and results leave through shard 0
 * (hydia_group_ctx(g, 0): encrypt / import the query there, decrypt there); result batches are in GLOBAL block order, so
 * hydia_decrypt_index returns database indices.  Results are bit-identical to one context holding the whole database. */
typedef struct hydia_group hydia_group;
/* block range [lo, hi) of `rank`: the first total_blocks % world ranks take one extra block (host only, no GPU needed) */
void hydia_shard_blocks(size_t total_blocks, uint32_t world, uint32_t rank, size_t *lo, size_t *hi);
/* a device index this node does not have: HYDIA_ERR_DEVICE, nothing is created (1 to 16 shards; a negative index: HYDIA_ERR_ARG) */
int hydia_group_create(const hydia_params *p, const int *devices /* [n_shards] GPU index of each shard */, uint32_t n_shards,
                       hydia_group **out);
void hydia_group_destroy(hydia_group *g);
uint32_t hydia_group_size(const hydia_group *g);
hydia_ctx *hydia_group_ctx(hydia_group *g, uint32_t shard); /* borrowed: never hydia_ctx_destroy it */
int hydia_group_keygen(hydia_group *g, const uint8_t seed[32]);
/* DiagonalEnroller::serializeDB over the group (normalises db IN PLACE); shard r enrols rows [first, first + n) of
 * hydia_group_shard_range */
int hydia_group_db_enroll(hydia_group *g, double *db /* n x vector_dim */, size_t n, const uint8_t seed[32]);
int hydia_group_shard_range(const hydia_group *g, uint32_t shard, size_t *first_vector, size_t *n_vectors);
/* How loop A (the 511 hoisted rotations) is shared: 0 = every shard computes all of them itself (nothing exchanged before the
 * mat-vec); 1 = shard k of the K active ones computes the contiguous range hydia_shard_blocks(vector_dim, K, k) and the ranges are
 * exchanged by peer copies (SURVEY 8e option B: loop A's work is done once per node instead of once per GPU).  Default 1.
 * Results are bit-identical either way. */
int hydia_group_set_rotation_split(hydia_group *g, int on);
/* Sender::computeSimilarity / indexScenario / membershipScenario over all shards; query and *out live in shard 0 */
int hydia_group_compute_similarity(hydia_group *g, const hydia_ct *query, hydia_ct **out);
int hydia_group_index_scenario(hydia_group *g, const hydia_ct *query, hydia_ct **out);
int hydia_group_membership_scenario(hydia_group *g, const hydia_ct *query, hydia_ct **out);

/* ---- HERS, approach 4: the paper's main comparison on the same kernels ---- */
/* HersEnroller::serializeDB, src/enroller/enroller_hers.cpp:40-93: index-batched (column) packing, vector_dim ciphertexts per
 * `slots`-vector matrix, normalises db IN PLACE; replaces the resident database */
int hydia_hers_db_enroll(hydia_ctx *ctx, double *db /* n x vector_dim */, size_t n, const uint8_t seed[32]);
/* HersReceiver::encryptQuery, src/receiver/receiver_hers.cpp:13-24: vector_dim ciphertexts, coordinate i in every slot */
int hydia_hers_encrypt_query(hydia_ctx *ctx, const double *query, const uint8_t seed[32], uint64_t nonce0, hydia_ct **out);
/* HersSender::computeSimilarity / indexScenario / membershipScenario, src/sender/sender_hers.cpp:13-58
 * (relinearise + rescale after every one of the vector_dim products of a block, :70-75) */
int hydia_hers_compute_similarity(hydia_ctx *ctx, const hydia_ct *query, hydia_ct **out);
int hydia_hers_index_scenario(hydia_ctx *ctx, const hydia_ct *query, hydia_ct **out);
int hydia_hers_membership_scenario(hydia_ctx *ctx, const hydia_ct *query, hydia_ct **out);

/* ---- evaluator primitives (used by the parity tests and by adapters) ---- */
int hydia_ntt(hydia_ctx *ctx, uint64_t *data /* host, [count][N] in place */, uint32_t count, uint32_t modulus_index,
              int inverse);
int hydia_eval_rotate(hydia_ctx *ctx, const hydia_ct *in, int rot, hydia_ct **out);
int hydia_eval_mult(hydia_ctx *ctx, const hydia_ct *a, const hydia_ct *b, hydia_ct **out); /* mult+relin+rescale */
int hydia_eval_mult_no_relin(hydia_ctx *ctx, const hydia_ct *a, const hydia_ct *b, hydia_ct **out);
int hydia_relinearize(hydia_ctx *ctx, hydia_ct *ct);
int hydia_rescale(hydia_ctx *ctx, hydia_ct *ct);
int hydia_eval_add(hydia_ctx *ctx, hydia_ct *a, const hydia_ct *b);
int hydia_level_reduce(hydia_ctx *ctx, hydia_ct *ct, uint32_t n_limbs);

/* ---- measurement: HIP-event time of named kernels on the context's stream since the last reset
 * ("hydia_tensor" = loop B's tensor-accumulate kernel, "ks_inner_product") ---- */
int hydia_kernel_time(hydia_ctx *ctx, const char *name, double *total_ms, uint64_t *launches);
int hydia_kernel_time_reset(hydia_ctx *ctx);
/* Byte ledger (process-wide): while enabled every kernel launcher records the bytes its launch has to move, by kernel name.
 * The current table ("kernel<TAB>launches<TAB>bytes" lines) is written to out (NUL-terminated, at most cap bytes; needed = full
 * size), THEN enable is applied: 1 = clear and record, 0 = stop and clear, -1 = leave as is.  tools/kernel_rooflines.py divides
 * by the rocprofv3 kernel times of the same run. */
int hydia_byte_ledger(int enable, char *out, size_t cap, size_t *needed);
/* NTT microbenchmark on pooled scratch memory: `polys` polynomials x moduli [first_mod, first_mod + n_mods), in place,
 * HIP-event milliseconds per iteration (tools/bench_ntt.py; 512 KiB algorithmic per limb-transform) */
int hydia_bench_ntt(hydia_ctx *ctx, uint32_t polys, uint32_t first_mod, uint32_t n_mods, int inverse, uint32_t iters,
                    double *ms_per_iter);

#ifdef __cplusplus
}
#endif
#endif
