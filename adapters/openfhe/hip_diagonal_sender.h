// adapters/openfhe/hip_diagonal_sender.h — reference-side binding of libhydia.so.
//
// STATUS: written against the OpenFHE v1.2.3 public API from memory; NOT compiled in this repository's image, where
// OpenFHE is absent.  Everything it calls on the hydia side is exercised by the test-suite with
// externally supplied prime chains, keys, ciphertexts and database (tests/test_gpu_parity.py::
// test_custom_prime_chain_context_bit_exact, test_gpu_client.py::test_full_size_2p20_properties step 4), so what remains
// unverified is only the OpenFHE half of each marshalling function.  Build inside the reference tree with
//     -DHYDIA_WITH_OPENFHE -I<hydia>/include -L<hydia>/image_matching_amd -lhydia
// and in src/main.cpp:324-327 replace `new DiagonalSender(cc, pk, numVectors)` by `new HipDiagonalSender(cc, pk, numVectors)`.
//
// What stays with OpenFHE: key generation, encryption, decryption, the enroller's file output.  What moves to the GPU:
// DiagonalSender::computeSimilarity / indexScenario / membershipScenario (src/sender/sender_diag.cpp:12-94) and
// OpenFHEWrapper::chebyshevCompare (src/openFHE_wrapper.cpp:143-185).
//
// Data crosses in COEFFICIENT form and is converted with hydia_ntt on the device: the library's evaluation order is tied to
// ITS 2N-th roots (include/hydia.h:10-15).  A context created with OpenFHE's own roots (third argument of
// hydia_ctx_create_custom) makes EVALUATION-form crossing possible if OpenFHE's bit-reversed layout is the same
// a(psi^(2*bitrev(j)+1)); that equality cannot be checked here, so the safe route is the default.
#pragma once
#ifdef HYDIA_WITH_OPENFHE

#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sender_diag.h"  // the reference's DiagonalSender (include/sender_diag.h:5-28)
#include "hydia.h"

namespace hydia_adapter {
using namespace lbcrypto;

inline void check(int code, const char *what) {
    if (code != 0) throw std::runtime_error(std::string(what) + ": " + hydia_last_error());
}

// one DCRTPoly -> [limb][N] residues in coefficient form
inline void limbs_coeff(DCRTPoly poly, std::vector<uint64_t> &out) {
    poly.SetFormat(Format::COEFFICIENT);
    for (size_t j = 0; j < poly.GetNumOfElements(); j++) {
        const auto &v = poly.GetElementAtIndex(j).GetValues();
        for (size_t i = 0; i < v.GetLength(); i++) out.push_back(v[i].ConvertToInt());
    }
}

class Bridge {
  public:
    hydia_ctx *hx = nullptr;
    hydia_info info{};
    CryptoContext<DCRTPoly> cc;

    Bridge(CryptoContext<DCRTPoly> ccParam, int device = 0) : cc(ccParam) {
        const auto cp = std::dynamic_pointer_cast<CryptoParametersCKKSRNS>(cc->GetCryptoParameters());
        std::vector<uint64_t> moduli;
        for (const auto &p : cp->GetElementParams()->GetParams()) moduli.push_back(p->GetModulus().ConvertToInt());
        const uint32_t nq = (uint32_t)moduli.size();
        for (const auto &p : cp->GetParamsP()->GetParams()) moduli.push_back(p->GetModulus().ConvertToInt());
        hydia_params prm;
        hydia_default_params(&prm);  // src/main.cpp:169-173; the chain itself comes from cc
        prm.log_n = 0;
        for (uint32_t n = cc->GetRingDimension(); n > 1; n >>= 1) prm.log_n++;
        prm.mult_depth = nq - 1;
        prm.dnum = cp->GetNumPartQ();
        check(hydia_ctx_create_custom(&prm, moduli.data(), /*roots*/ nullptr, nq, (uint32_t)moduli.size() - nq, device, &hx),
              "hydia_ctx_create_custom");
        hydia_get_info(hx, &info);
        if (info.alpha != cp->GetNumPerPartQ()) throw std::runtime_error("hydia adapter: digit partition differs from OpenFHE's");
    }
    ~Bridge() { hydia_ctx_destroy(hx); }

    // [count][N] coefficient-form residues of modulus m -> evaluation form, in place (device NTT)
    void to_eval(uint64_t *data, uint32_t count, uint32_t m) { check(hydia_ntt(hx, data, count, m, 0), "hydia_ntt"); }
    void to_coeff(uint64_t *data, uint32_t count, uint32_t m) { check(hydia_ntt(hx, data, count, m, 1), "hydia_ntt"); }

    // EvalKeyRelin (hybrid): for digit d, (b_d, a_d) over Q u P  ->  [dnum][2][n_q+n_p][N], include/hydia.h:16
    void import_eval_key(const EvalKey<DCRTPoly> &key, int rot) {
        const auto &B = key->GetBVector(), &A = key->GetAVector();
        const size_t nt = info.n_q + info.n_p, N = info.n;
        std::vector<uint64_t> buf;
        buf.reserve(B.size() * 2 * nt * N);
        for (size_t d = 0; d < B.size(); d++) {
            limbs_coeff(B[d], buf);
            limbs_coeff(A[d], buf);
        }
        for (size_t d = 0; d < B.size() * 2; d++)
            for (size_t m = 0; m < nt; m++) to_eval(buf.data() + (d * nt + m) * N, 1, (uint32_t)m);
        check(hydia_import_eval_key(hx, rot, buf.data()), "hydia_import_eval_key");
    }
    // relinearisation key + rotation keys 1..511 and the powers of two EvalSum needs (src/main.cpp:184-206)
    void import_keys(const std::string &keyTag) {
        import_eval_key(cc->GetEvalMultKeyVector(keyTag)[0], 0);
        const auto &amap = cc->GetEvalAutomorphismKeyMap(keyTag);
        std::vector<int> rots;
        for (int r = 1; r < (int)info.vector_dim; r++) rots.push_back(r);
        for (int r = (int)info.vector_dim; r <= (int)info.slots / 2; r <<= 1) rots.push_back(r);
        for (int r : rots) {
            const auto it = amap.find(cc->FindAutomorphismIndex(r));
            if (it == amap.end()) throw std::runtime_error("hydia adapter: missing rotation key " + std::to_string(r));
            import_eval_key(it->second, r);
        }
    }
    // a fresh 2-component ciphertext at level 0 -> [2][n_q][N] evaluation form
    std::vector<uint64_t> marshal(const Ciphertext<DCRTPoly> &ct) {
        std::vector<uint64_t> buf;
        for (const auto &e : ct->GetElements()) limbs_coeff(e, buf);
        const size_t nl = ct->GetElements()[0].GetNumOfElements(), N = info.n;
        for (size_t p = 0; p < 2; p++)
            for (size_t m = 0; m < nl; m++) to_eval(buf.data() + (p * nl + m) * N, 1, (uint32_t)m);
        return buf;
    }
    // serial/db_diagonal/index<t>.bin (enroller_diag.cpp:161) read ONCE into HBM instead of per query (sender_diag.cpp:87-91)
    void load_database(size_t numVectors) {
        check(hydia_db_alloc(hx, numVectors), "hydia_db_alloc");
        const size_t cts = hydia_db_num_cts(hx, numVectors);
        for (size_t t = 0; t < cts; t++) {
            Ciphertext<DCRTPoly> ct;
            if (!Serial::DeserializeFromFile("serial/db_diagonal/index" + std::to_string(t) + ".bin", ct, SerType::BINARY))
                throw std::runtime_error("hydia adapter: cannot read database ciphertext " + std::to_string(t));
            check(hydia_db_import_ct(hx, t, marshal(ct).data()), "hydia_db_import_ct");
        }
    }
    // device batch -> OpenFHE ciphertexts shaped like `like` (metadata: level = dropped limbs, noiseScaleDeg 1)
    std::vector<Ciphertext<DCRTPoly>> unmarshal(hydia_ct *h, const Ciphertext<DCRTPoly> &like) {
        uint32_t count, npoly, nl;
        double scale;
        hydia_ct_shape(h, &count, &npoly, &nl, &scale);
        const size_t N = info.n;
        std::vector<uint64_t> buf((size_t)count * npoly * nl * N);
        check(hydia_ct_export(hx, h, buf.data()), "hydia_ct_export");
        std::vector<Ciphertext<DCRTPoly>> out;
        for (uint32_t x = 0; x < count; x++) {
            auto ct = like->Clone();
            std::vector<DCRTPoly> elems;
            for (uint32_t p = 0; p < npoly; p++) {
                DCRTPoly poly(like->GetElements()[0]);
                poly.DropLastElements(poly.GetNumOfElements() - nl);
                poly.SetFormat(Format::COEFFICIENT);
                for (uint32_t m = 0; m < nl; m++) {
                    uint64_t *src = buf.data() + (((size_t)x * npoly + p) * nl + m) * N;
                    to_coeff(src, 1, m);
                    NativePoly limb = poly.GetElementAtIndex(m);
                    NativeVector v(N, limb.GetModulus());
                    for (size_t i = 0; i < N; i++) v[i] = NativeInteger(src[i]);
                    limb.SetValues(v, Format::COEFFICIENT);
                    poly.SetElementAtIndex(m, limb);
                }
                poly.SetFormat(Format::EVALUATION);
                elems.push_back(poly);
            }
            ct->SetElements(elems);
            ct->SetLevel(info.n_q - nl);
            ct->SetNoiseScaleDeg(1);
            ct->SetScalingFactor(scale);
            out.push_back(ct);
        }
        return out;
    }
};

// The reference's DiagonalSender with its three scenario methods served by the GPU.
class HipDiagonalSender : public DiagonalSender {
  public:
    HipDiagonalSender(CryptoContext<DCRTPoly> ccParam, PublicKey<DCRTPoly> pkParam, size_t vectorParam, int device = 0)
        : DiagonalSender(ccParam, pkParam, vectorParam), bridge(ccParam, device), n(vectorParam) {
        bridge.import_keys(pkParam->GetKeyTag());
        bridge.load_database(n);
    }
    std::vector<Ciphertext<DCRTPoly>> computeSimilarity(std::vector<Ciphertext<DCRTPoly>> &queryCipher) override {
        return run(queryCipher, hydia_compute_similarity);
    }
    std::vector<Ciphertext<DCRTPoly>> indexScenario(std::vector<Ciphertext<DCRTPoly>> &queryCipher) override {
        return run(queryCipher, hydia_index_scenario);
    }
    Ciphertext<DCRTPoly> membershipScenario(std::vector<Ciphertext<DCRTPoly>> &queryCipher) override {
        return run(queryCipher, hydia_membership_scenario)[0];
    }

  private:
    Bridge bridge;
    size_t n;
    std::vector<Ciphertext<DCRTPoly>> run(std::vector<Ciphertext<DCRTPoly>> &queryCipher,
                                          int (*fn)(hydia_ctx *, const hydia_ct *, hydia_ct **)) {
        hydia_ct *q = nullptr, *out = nullptr;
        const auto buf = bridge.marshal(queryCipher[0]);
        check(hydia_ct_import(bridge.hx, buf.data(), 1, 2, bridge.info.n_q, queryCipher[0]->GetScalingFactor(), &q), "hydia_ct_import");
        const int rc = fn(bridge.hx, q, &out);
        hydia_ct_free(q);
        check(rc, "sender scenario");
        auto res = bridge.unmarshal(out, queryCipher[0]);
        hydia_ct_free(out);
        return res;
    }
};

}  // namespace hydia_adapter
#endif  // HYDIA_WITH_OPENFHE
