// image_matching_amd/csrc/client.h — receiver / enroller / key-generation engines (GPU): sampling, canonical-embedding
// encode/decode, public-key encryption, decryption, on-GPU enrolment into the HBM-resident diagonal layout.
#pragma once
#include "hydia_core.h"

namespace hydia {
void client_keygen(Context &cx, const uint8_t seed[32]);
Ct client_encrypt(Context &cx, const double *slots, int count, const uint8_t seed[32], uint64_t nonce0);
Ct client_encrypt_query(Context &cx, const double *query, const uint8_t seed[32], uint64_t nonce);
void client_decrypt(Context &cx, const Ct &ct, double *out);
// first_block: index of this context's first 16384-vector block inside the whole database (a shard of a multi-GPU database
// encrypts with the nonces the unsharded enrolment would use, so shards hold bit-identical ciphertexts)
// babies < vector_dim: diagonals pre-rotated for the baby-step / giant-step mat-vec with that many hoisted rotations
// (Context::similarity_bsgs_sum); same ciphertext order and nonces.  0 or vector_dim: the reference's layout
void client_enroll(Context &cx, double *db, size_t n, const uint8_t seed[32], size_t first_block = 0, int babies = 0);
// HERS (approach 4): column-packed enrolment and the vector_dim broadcast query ciphertexts
void client_hers_enroll(Context &cx, double *db, size_t n, const uint8_t seed[32]);
Ct client_hers_encrypt_query(Context &cx, const double *query, const uint8_t seed[32], uint64_t nonce0);
}  // namespace hydia
