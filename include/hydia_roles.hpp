// include/hydia_roles.hpp — the reference's C++ role surface for approach 5 (HyDia) and approach 4 (HERS), over the C-ABI of hydia.h.
//
// Same class and method names as /root/reference/include/{sender,sender_diag,receiver,receiver_hers,receiver_diag,
// enroller_diag,enroller_hers}.h and the same constructor arguments (include/sender.h:22 `(cc, pk, numVectors)`,
// include/receiver.h:20-21 `(cc, pk, sk, numVectors)`, include/enroller_hers.h:19), so that src/main.cpp's cases 4 and 5
// (:183-192, :243-247, :319-327, :333-374) read unchanged; the OpenFHE handle types are replaced by thin handles onto HBM-resident
// objects:
//     CryptoContext<DCRTPoly>            ->  hydia::CryptoContext (context + keys + resident database)
//     PublicKey / PrivateKey<DCRTPoly>   ->  hydia::PublicKey / PrivateKey (placeholders: the key material lives in the context)
//     Ciphertext<DCRTPoly>               ->  hydia::Ciphertext   (one element of a device batch)
// `using namespace hydia::ofhe;` gives these the reference's template spelling (Ciphertext<DCRTPoly>, ...).
// Randomness: like the reference (OpenFHE seeds its PRNG from the OS) every role object draws its 32-byte sampler key from the
// operating system (hydia_random_seed) unless the caller passes one for reproducibility; nonces count up per object.
// Error behaviour mirrors the reference: a message on cerr and carry on (src/sender/sender_diag.cpp:89-91); the
// status code of the last failing call is kept in CryptoContext::last_status for callers that want to assert.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "hydia.h"

namespace hydia {

const double MATCH_THRESHOLD = 0.44;  // include/config.h:9
const size_t COMP_DEPTH = 10;         // include/config.h:14
const size_t VECTOR_DIM = 512;        // include/config.h:30

inline void role_seed(uint8_t out[32], const uint8_t *seed32) {
    if (seed32) {
        for (int i = 0; i < 32; i++) out[i] = seed32[i];
    } else if (hydia_random_seed(out) != 0) {
        std::cerr << "Error: " << hydia_last_error() << std::endl;
        std::abort();  // never encrypt under a predictable key
    }
}

class CryptoContextImpl;
// Placeholders for OpenFHE's PublicKey<DCRTPoly> / PrivateKey<DCRTPoly>: the key material stays inside the context (HBM), the
// handles only say "this context's keys" so that the reference's constructor calls keep their shape.
struct PublicKey {
    const CryptoContextImpl *owner = nullptr;
    explicit operator bool() const { return owner != nullptr; }
};
struct PrivateKey {
    const CryptoContextImpl *owner = nullptr;
    explicit operator bool() const { return owner != nullptr; }
};
struct KeyPair {  // what cc->KeyGen() returns (src/main.cpp:183-185)
    PublicKey publicKey;
    PrivateKey secretKey;
    bool good() const { return (bool)publicKey && (bool)secretKey; }
    explicit operator bool() const { return good(); }
};

class CryptoContextImpl {
  public:
    hydia_ctx *h = nullptr;
    hydia_info info{};
    int last_status = 0;
    hydia_group *group = nullptr;  // set when the context is sharded over several GPUs; h is then the group's shard 0
    explicit CryptoContextImpl(const hydia_params &p, int device = 0) {
        last_status = hydia_ctx_create(&p, device, &h);
        if (last_status != 0) {
            std::cerr << "Error: " << hydia_last_error() << std::endl;
            h = nullptr;
            return;
        }
        hydia_get_info(h, &info);
    }
    // one context per entry of `devices` (DB row-blocks sharded across them); queries and results use shard 0
    CryptoContextImpl(const hydia_params &p, const std::vector<int> &devices) {
        last_status = hydia_group_create(&p, devices.data(), (uint32_t)devices.size(), &group);
        if (last_status != 0) {
            std::cerr << "Error: " << hydia_last_error() << std::endl;
            group = nullptr;
            return;
        }
        h = hydia_group_ctx(group, 0);
        hydia_get_info(h, &info);
    }
    ~CryptoContextImpl() {
        if (group) hydia_group_destroy(group);
        else hydia_ctx_destroy(h);
    }
    CryptoContextImpl(const CryptoContextImpl &) = delete;
    CryptoContextImpl &operator=(const CryptoContextImpl &) = delete;
    bool check(int code, const char *what) {
        if (code != 0) {
            last_status = code;
            std::cerr << "Error: " << what << ": " << hydia_last_error() << std::endl;
        }
        return code == 0;
    }
    size_t GetRingDimension() const { return info.n; }
    size_t GetBatchSize() const { return info.slots; }
    // cc->KeyGen() (src/main.cpp:183): secret, public, relinearisation, rotation {1..dim-1, dim*2^k} keys in ONE call — everything
    // EvalMultKeyGen / EvalSumKeyGen / EvalRotateKeyGen (:187-206) would add; seed32 == nullptr draws the key material from the OS.
    // The returned pair is empty (good() == false) when generation failed.
    KeyPair KeyGen(const uint8_t *seed32 = nullptr) {
        uint8_t seed[32];
        role_seed(seed, seed32);
        if (!check(group ? hydia_group_keygen(group, seed) : hydia_keygen(h, seed), "key generation")) return KeyPair{};
        return KeyPair{PublicKey{this}, PrivateKey{this}};
    }
    // src/main.cpp:187-206: the evaluation keys exist since KeyGen; these keep the reference's call sequence compiling
    void EvalMultKeyGen(const PrivateKey &) {}
    void EvalSumKeyGen(const PrivateKey &) {}
    template <class IndexList>
    void EvalRotateKeyGen(const PrivateKey &, const IndexList &) {}
};
using CryptoContext = std::shared_ptr<CryptoContextImpl>;

inline CryptoContext GenCryptoContext(size_t multDepth = 11, uint32_t scalingModSize = 45, uint32_t vectorDim = 512,
                                      uint32_t logN = 15, int device = 0) {
    hydia_params p;
    hydia_default_params(&p);
    p.mult_depth = (uint32_t)multDepth;
    p.scale_bits = scalingModSize;
    p.vector_dim = vectorDim;
    p.log_n = logN;
    return std::make_shared<CryptoContextImpl>(p, device);
}
// the same context sharded over several GPUs (or several shards on one): DiagonalEnroller / DiagonalSender then work on
// the whole group, each GPU owning a contiguous range of 16384-vector blocks
inline CryptoContext GenShardedCryptoContext(const std::vector<int> &devices, size_t multDepth = 11, uint32_t scalingModSize = 45,
                                             uint32_t vectorDim = 512, uint32_t logN = 15) {
    hydia_params p;
    hydia_default_params(&p);
    p.mult_depth = (uint32_t)multDepth;
    p.scale_bits = scalingModSize;
    p.vector_dim = vectorDim;
    p.log_n = logN;
    return std::make_shared<CryptoContextImpl>(p, devices);
}

// one ciphertext = (shared device batch, index inside it)
struct CtBatch {
    CryptoContext cc;
    hydia_ct *h = nullptr;
    CtBatch(CryptoContext c, hydia_ct *p) : cc(std::move(c)), h(p) {}
    ~CtBatch() { hydia_ct_free(h); }
    uint32_t count() const {
        uint32_t c = 0;
        if (h) hydia_ct_shape(h, &c, nullptr, nullptr, nullptr);
        return c;
    }
};
struct Ciphertext {
    std::shared_ptr<CtBatch> batch;
    uint32_t index = 0;
    explicit operator bool() const { return batch && batch->h; }
};
inline std::vector<Ciphertext> split_batch(const CryptoContext &cc, hydia_ct *h) {
    std::vector<Ciphertext> v;
    if (!h) return v;
    auto b = std::make_shared<CtBatch>(cc, h);
    for (uint32_t i = 0; i < b->count(); i++) v.push_back(Ciphertext{b, i});
    return v;
}

// the reference's template spelling of the handle types: `using namespace hydia::ofhe;` in place of `using namespace lbcrypto;`
namespace ofhe {
struct DCRTPoly {};
template <class Element>
using CryptoContext = ::hydia::CryptoContext;
template <class Element>
using Ciphertext = ::hydia::Ciphertext;
template <class Element>
using PublicKey = ::hydia::PublicKey;
template <class Element>
using PrivateKey = ::hydia::PrivateKey;
}  // namespace ofhe

namespace OpenFHEWrapper {
// src/openFHE_wrapper.cpp:6-44
inline size_t computeRequiredDepth(size_t approach) { return hydia_compute_required_depth(approach); }
// src/openFHE_wrapper.cpp:81-85 (whole batch the ciphertext belongs to; returns the slots of ct.index)
inline std::vector<double> decryptToVector(CryptoContext cc, Ciphertext ctxt) {
    std::vector<double> all((size_t)ctxt.batch->count() * cc->info.slots);
    cc->check(hydia_decrypt(cc->h, ctxt.batch->h, all.data()), "decrypt");
    return std::vector<double>(all.begin() + (size_t)ctxt.index * cc->info.slots,
                               all.begin() + (size_t)(ctxt.index + 1) * cc->info.slots);
}
}  // namespace OpenFHEWrapper

// ---- include/sender.h:19-43
class Sender {
  public:
    Sender(CryptoContext ccParam, size_t vectorParam) : cc(std::move(ccParam)), numVectors(vectorParam) {}
    Sender(CryptoContext ccParam, PublicKey pkParam, size_t vectorParam) : cc(std::move(ccParam)), pk(pkParam), numVectors(vectorParam) {}  // include/sender.h:22
    virtual ~Sender() = default;
    virtual std::vector<Ciphertext> computeSimilarity(std::vector<Ciphertext> &queryCipher) = 0;
    virtual Ciphertext membershipScenario(std::vector<Ciphertext> &queryCipher) = 0;
    virtual std::vector<Ciphertext> indexScenario(std::vector<Ciphertext> &queryCipher) = 0;

  protected:
    CryptoContext cc;
    PublicKey pk;
    size_t numVectors;
};
// ---- include/sender_diag.h:5-28 (HersSender, approach 4, is further down).  On a sharded context (GenShardedCryptoContext)
// the same three methods run over every GPU of the group: per-shard mat-vec, results back in global block order, membership
// as per-shard EvalAddMany -> integer sum -> mod q -> EvalSum (hydia.h, "sharded sender").
class DiagonalSender : public Sender {
  public:
    DiagonalSender(CryptoContext ccParam, size_t vectorParam) : Sender(std::move(ccParam), vectorParam) {}
    DiagonalSender(CryptoContext ccParam, PublicKey pkParam, size_t vectorParam) : Sender(std::move(ccParam), pkParam, vectorParam) {}
    std::vector<Ciphertext> computeSimilarity(std::vector<Ciphertext> &queryCipher) override {
        hydia_ct *out = run(queryCipher, hydia_compute_similarity, hydia_group_compute_similarity, "computeSimilarity");
        return out ? split_batch(cc, out) : std::vector<Ciphertext>{};
    }
    Ciphertext membershipScenario(std::vector<Ciphertext> &queryCipher) override {
        hydia_ct *out = run(queryCipher, hydia_membership_scenario, hydia_group_membership_scenario, "membershipScenario");
        return out ? split_batch(cc, out)[0] : Ciphertext{};
    }
    std::vector<Ciphertext> indexScenario(std::vector<Ciphertext> &queryCipher) override {
        hydia_ct *out = run(queryCipher, hydia_index_scenario, hydia_group_index_scenario, "indexScenario");
        return out ? split_batch(cc, out) : std::vector<Ciphertext>{};
    }

  private:
    hydia_ct *run(std::vector<Ciphertext> &q, int (*one)(hydia_ctx *, const hydia_ct *, hydia_ct **),
                  int (*sharded)(hydia_group *, const hydia_ct *, hydia_ct **), const char *what) {
        if (q.empty() || !q[0]) {
            std::cerr << "Error: empty query ciphertext" << std::endl;
            return nullptr;
        }
        hydia_ct *out = nullptr;
        const int code = cc->group ? sharded(cc->group, q[0].batch->h, &out) : one(cc->h, q[0].batch->h, &out);
        return cc->check(code, what) ? out : nullptr;
    }
};

// ---- include/receiver.h:17-43, include/receiver_hers.h, include/receiver_diag.h
class Receiver {
  public:
    Receiver(CryptoContext ccParam, size_t vectorParam, const uint8_t *seed32 = nullptr) : cc(std::move(ccParam)), numVectors(vectorParam) {
        role_seed(seed, seed32);
    }
    Receiver(CryptoContext ccParam, PublicKey pkParam, PrivateKey skParam, size_t vectorParam)  // include/receiver.h:20-21
        : cc(std::move(ccParam)), pk(pkParam), sk(skParam), numVectors(vectorParam) {
        role_seed(seed, nullptr);
    }
    virtual ~Receiver() = default;
    virtual std::vector<Ciphertext> encryptQuery(std::vector<double> query) = 0;
    virtual bool decryptMembership(Ciphertext &membershipCipher) = 0;
    virtual std::vector<size_t> decryptIndex(std::vector<Ciphertext> &indexCipher) = 0;

  protected:
    CryptoContext cc;
    PublicKey pk;
    PrivateKey sk;
    size_t numVectors;
    uint8_t seed[32];    // this object's sampler key (OS entropy unless supplied); nonces count up per object
    uint64_t nonce = 0;
};
// approach 4's receiver (include/receiver_hers.h:9-28); DiagonalReceiver inherits its decrypt* and overrides encryptQuery
class HersReceiver : public Receiver {
  public:
    using Receiver::Receiver;
    // src/receiver/receiver_hers.cpp:13-24: vector_dim ciphertexts, one per dimension
    std::vector<Ciphertext> encryptQuery(std::vector<double> query) override {
        hydia_ct *out = nullptr;
        if (query.size() < cc->info.vector_dim) query.resize(cc->info.vector_dim, 0.0);
        nonce += cc->info.vector_dim;
        if (!cc->check(hydia_hers_encrypt_query(cc->h, query.data(), seed, nonce, &out), "encryptQuery")) return {};
        return split_batch(cc, out);
    }
    // src/receiver/receiver_hers.cpp:26-35
    bool decryptMembership(Ciphertext &membershipCipher) override {
        if (!membershipCipher) return false;
        return OpenFHEWrapper::decryptToVector(cc, membershipCipher)[0] >= 1.0;
    }
    // src/receiver/receiver_hers.cpp:37-54
    std::vector<size_t> decryptIndex(std::vector<Ciphertext> &indexCipher) override {
        size_t batchSize = cc->GetBatchSize();
        std::vector<size_t> outputValues;
        for (size_t i = 0; i < indexCipher.size(); i++) {
            if (!indexCipher[i]) continue;
            std::vector<double> indexValues = OpenFHEWrapper::decryptToVector(cc, indexCipher[i]);
            for (size_t j = 0; j < batchSize; j++)
                if (indexValues[j] >= 1.0) outputValues.push_back(j + (i * batchSize));
        }
        return outputValues;
    }
};
using HersQueryReceiver = HersReceiver;  // round-2 name of the approach-4 receiver
class DiagonalReceiver : public HersReceiver {
  public:
    using HersReceiver::HersReceiver;
    // src/receiver/receiver_diag.cpp:13-26
    std::vector<Ciphertext> encryptQuery(std::vector<double> query) override {
        hydia_ct *out = nullptr;
        if (query.size() < cc->info.vector_dim) query.resize(cc->info.vector_dim, 0.0);
        if (!cc->check(hydia_encrypt_query(cc->h, query.data(), seed, ++nonce, &out), "encryptQuery")) return {};
        return split_batch(cc, out);
    }
};

// ---- enrollers.  Randomness: without a caller-supplied seed EVERY serializeDB call draws a fresh sampler key from the OS (the
// database nonces restart at the same base on every call, so a key must never serve two enrolments: ct2 - ct1 would be the
// plaintext difference); with a supplied seed (reproducible tests) a second enrolment on the same object is refused.
class EnrollerBase {
  protected:
    EnrollerBase(CryptoContext ccParam, PublicKey pkParam, size_t vectorParam, const uint8_t *seed32)
        : cc(std::move(ccParam)), pk(pkParam), numVectors(vectorParam), own_seed(seed32 == nullptr) {
        if (seed32) role_seed(seed, seed32);
    }
    bool next_seed(const char *what) {
        if (own_seed) {
            role_seed(seed, nullptr);
        } else if (enrolled) {
            cc->last_status = HYDIA_ERR_STATE;
            std::cerr << "Error: " << what << ": a caller-supplied seed enrols ONE database; construct a new enroller" << std::endl;
            return false;
        }
        enrolled = true;
        return true;
    }
    std::vector<double> flatten(const std::vector<std::vector<double>> &database) const {
        const size_t dim = cc->info.vector_dim;
        std::vector<double> flat(numVectors * dim, 0.0);
        for (size_t i = 0; i < numVectors && i < database.size(); i++)
            for (size_t j = 0; j < dim && j < database[i].size(); j++) flat[i * dim + j] = database[i][j];
        return flat;
    }
    void write_back(const std::vector<double> &flat, std::vector<std::vector<double>> &database) const {  // normalised in place
        const size_t dim = cc->info.vector_dim;
        for (size_t i = 0; i < numVectors && i < database.size(); i++)
            for (size_t j = 0; j < dim && j < database[i].size(); j++) database[i][j] = flat[i * dim + j];
    }
    CryptoContext cc;
    PublicKey pk;
    size_t numVectors;
    uint8_t seed[32] = {};
    bool own_seed, enrolled = false;
};
// ---- include/enroller_diag.h:7-27
class DiagonalEnroller : public EnrollerBase {
  public:
    DiagonalEnroller(CryptoContext ccParam, size_t vectorParam, const uint8_t *seed32 = nullptr)
        : EnrollerBase(std::move(ccParam), PublicKey{}, vectorParam, seed32) {}
    DiagonalEnroller(CryptoContext ccParam, PublicKey pkParam, size_t vectorParam)  // src/main.cpp:246
        : EnrollerBase(std::move(ccParam), pkParam, vectorParam, nullptr) {}
    // src/enroller/enroller_diag.cpp:12-53 — normalises `database` in place; ciphertexts go to HBM, not to
    // serial/db_diagonal/index<t>.bin
    void serializeDB(std::vector<std::vector<double>> &database) {
        if (!next_seed("serializeDB")) return;
        std::vector<double> flat = flatten(database);
        if (!cc->check(cc->group ? hydia_group_db_enroll(cc->group, flat.data(), numVectors, seed)
                                 : hydia_db_enroll(cc->h, flat.data(), numVectors, seed),
                       "serializeDB"))
            return;
        write_back(flat, database);
    }
};

// ---- HERS, approach 4: include/sender_hers.h:9-44, include/receiver_hers.h:9-28, include/enroller_hers.h:16-37.
// The query is vector_dim ciphertexts (one batch handle, split per element like the reference's vector).
class HersSender : public Sender {
  public:
    HersSender(CryptoContext ccParam, size_t vectorParam) : Sender(std::move(ccParam), vectorParam) {}
    HersSender(CryptoContext ccParam, PublicKey pkParam, size_t vectorParam) : Sender(std::move(ccParam), pkParam, vectorParam) {}
    std::vector<Ciphertext> computeSimilarity(std::vector<Ciphertext> &queryCipher) override {
        hydia_ct *out = nullptr;
        if (queryCipher.empty() || !queryCipher[0] ||
            !cc->check(hydia_hers_compute_similarity(cc->h, queryCipher[0].batch->h, &out), "computeSimilarity"))
            return {};
        return split_batch(cc, out);
    }
    Ciphertext membershipScenario(std::vector<Ciphertext> &queryCipher) override {
        hydia_ct *out = nullptr;
        if (queryCipher.empty() || !queryCipher[0] ||
            !cc->check(hydia_hers_membership_scenario(cc->h, queryCipher[0].batch->h, &out), "membershipScenario"))
            return Ciphertext{};
        return split_batch(cc, out)[0];
    }
    std::vector<Ciphertext> indexScenario(std::vector<Ciphertext> &queryCipher) override {
        hydia_ct *out = nullptr;
        if (queryCipher.empty() || !queryCipher[0] ||
            !cc->check(hydia_hers_index_scenario(cc->h, queryCipher[0].batch->h, &out), "indexScenario"))
            return {};
        return split_batch(cc, out);
    }
};
class HersEnroller : public EnrollerBase {  // include/enroller_hers.h:16-37
  public:
    HersEnroller(CryptoContext ccParam, size_t vectorParam, const uint8_t *seed32 = nullptr)
        : EnrollerBase(std::move(ccParam), PublicKey{}, vectorParam, seed32) {}
    HersEnroller(CryptoContext ccParam, PublicKey pkParam, size_t vectorParam)  // src/main.cpp:243
        : EnrollerBase(std::move(ccParam), pkParam, vectorParam, nullptr) {}
    void serializeDB(std::vector<std::vector<double>> &database) {  // enroller_hers.cpp:40-93
        if (!next_seed("serializeDB")) return;
        std::vector<double> flat = flatten(database);
        if (!cc->check(hydia_hers_db_enroll(cc->h, flat.data(), numVectors, seed), "serializeDB")) return;
        write_back(flat, database);
    }
};

}  // namespace hydia
