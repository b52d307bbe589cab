// image_matching_amd/csrc/cli_main.cpp — `ImageMatching <file.dat> <approach>` on the MI355X stack.
//
// Behaviour contract (what scripts around the reference's latency driver rely on): the stdout lines of
// /root/reference/src/main.cpp (:42, :101-107, :210, :290, :331-398) and one appended row of latency.csv with the twelve
// columns of /root/reference/tools/setup_experiment.sh:4-16.  The program itself is organised differently: the dataset is
// one value object, an approach is a small table entry that builds its three roles, and the five measured phases are a table
// of (stdout label, action) walked by one stopwatch — each action reports the CSV cells it owns.
//
//   HYDIA_DEVICES=0,1,2,3   shard the encrypted database over these GPUs (an index may repeat: several shards on one GPU)
//   HYDIA_SEED=<integer>    reproducible key / encryption randomness (default: operating-system entropy)
// Approaches 5 (HyDia) and 4 (HERS) exist on this stack; 1-3 are refused.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <memory>
#include <sstream>

#include "../../include/hydia_roles.hpp"

namespace {

using hydia::Ciphertext;
using hydia::CryptoContext;

struct Dataset {  // test/*.dat: n, the query (VECTOR_DIM values), then n rows
    size_t n = 0;
    std::vector<double> query;
    std::vector<std::vector<double>> rows;
    void read_body(std::istream &in) {
        query.assign(hydia::VECTOR_DIM, 0.0);
        for (double &v : query) in >> v;
        std::cout << "Reading database vectors from file... " << std::endl;
        rows.assign(n, std::vector<double>(hydia::VECTOR_DIM));
        for (auto &r : rows)
            for (double &v : r) in >> v;
    }
};

struct Roles {
    std::unique_ptr<hydia::Receiver> receiver;
    std::unique_ptr<hydia::Sender> sender;
};
struct Approach {
    size_t id;
    const char *banner, *csv_tag;
    std::function<void(CryptoContext, Dataset &, const uint8_t *)> enroll;
    std::function<Roles(CryptoContext, size_t, const uint8_t *)> make_roles;
};
const Approach APPROACHES[] = {
    {4, "Experimental approach: HERS paper", "HERS",
     [](CryptoContext cc, Dataset &d, const uint8_t *s) { hydia::HersEnroller(cc, d.n, s).serializeDB(d.rows); },
     [](CryptoContext cc, size_t n, const uint8_t *s) {
         return Roles{std::make_unique<hydia::HersQueryReceiver>(cc, n, s), std::make_unique<hydia::HersSender>(cc, n)};
     }},
    {5, "Experimental approach: Novel diagonal transform", "Diagonal",
     [](CryptoContext cc, Dataset &d, const uint8_t *s) { hydia::DiagonalEnroller(cc, d.n, s).serializeDB(d.rows); },
     [](CryptoContext cc, size_t n, const uint8_t *s) {
         return Roles{std::make_unique<hydia::DiagonalReceiver>(cc, n, s), std::make_unique<hydia::DiagonalSender>(cc, n)};
     }},
};

std::string show(const std::vector<size_t> &v) {  // OpenFHE's operator<< for vectors: "[ a b c ]"
    std::ostringstream os;
    os << "[ ";
    for (size_t x : v) os << x << " ";
    os << "]";
    return os.str();
}
std::vector<int> devices_from_env() {
    std::vector<int> d;
    if (const char *e = std::getenv("HYDIA_DEVICES")) {
        std::stringstream ss(e);
        for (std::string tok; std::getline(ss, tok, ',');)
            if (!tok.empty()) d.push_back(std::atoi(tok.c_str()));
    }
    return d;
}
// three independent 32-byte keys (key generation, enrolment, queries) from HYDIA_SEED, or nullptrs = OS entropy
struct Seeds {
    uint8_t keygen[32], enroll[32], query[32];
    bool fixed = false;
    Seeds() {
        const char *e = std::getenv("HYDIA_SEED");
        if (!e) return;
        fixed = true;
        const unsigned long long v = std::strtoull(e, nullptr, 0);
        uint8_t *out[3] = {keygen, enroll, query};
        for (int k = 0; k < 3; k++) {
            std::memset(out[k], 0, 32);
            std::memcpy(out[k], &v, sizeof v);
            out[k][31] = (uint8_t)(k + 1);
        }
    }
    const uint8_t *get(const uint8_t *s) const { return fixed ? s : nullptr; }
};

int usage_error(const char *msg) {
    std::cerr << "Error: " << msg << std::endl;
    return 1;
}

}  // namespace

int main(int argc, char *argv[]) {
    std::cout << "\tRunning Setup Operations:" << std::endl;
    if (argc < 2) return usage_error("input file not included");
    std::ifstream in(argv[1]);
    if (!in.is_open()) return usage_error("unable to open input file");
    Dataset data;
    in >> data.n;
    if (argc < 3) return usage_error("approach argument not included");
    const size_t wanted = (size_t)std::atoi(argv[2]);
    if (wanted < 1 || wanted > 5) return usage_error("approach must be from 1 to 5");
    const Approach *approach = nullptr;
    for (const Approach &a : APPROACHES)
        if (a.id == wanted) approach = &a;
    if (!approach)
        return usage_error("only approach 5 (novel diagonal transform, HyDia) and approach 4 (HERS) are built in hydia-mi355x");
    std::ofstream csv("latency.csv" /* EXP_FILEPATH, include/config.h:36 */, std::ios::app);
    if (!csv.is_open()) return usage_error("experiment file not found");

    const size_t depth = hydia::OpenFHEWrapper::computeRequiredDepth(approach->id);
    std::cout << approach->banner << std::endl;
    csv << approach->csv_tag << "," << std::flush;

    const Seeds seeds;
    const std::vector<int> devices = devices_from_env();
    CryptoContext cc = devices.empty() ? hydia::GenCryptoContext(depth, 45, hydia::VECTOR_DIM)
                                       : hydia::GenShardedCryptoContext(devices, depth, 45, hydia::VECTOR_DIM);
    if (!cc->h) return 2;
    if (!devices.empty() && approach->id != 5) return usage_error("HYDIA_DEVICES shards approach 5 only");
    std::cout << "Generating key pair, mult keys, sum keys and rotation keys on the GPU... " << std::endl;
    if (!cc->KeyGen(seeds.get(seeds.keygen))) return 2;
    std::cout << "CKKS scheme set up (depth = " << depth << ", batch size = " << cc->GetBatchSize() << ")" << std::endl;
    csv << data.n << "," << std::flush;

    data.read_body(in);
    in.close();
    std::cout << "Encrypting database vectors... " << std::endl;
    approach->enroll(cc, data, seeds.get(seeds.enroll));
    Roles roles = approach->make_roles(cc, data.n, seeds.get(seeds.query));

    // ---- the five measured phases
    std::vector<Ciphertext> query, index;
    Ciphertext membership;
    bool is_member = false;
    std::vector<size_t> hits;
    struct Phase {
        const char *label;
        std::function<std::string()> run;  // returns the CSV cell that follows the duration ("" = none)
    };
    const Phase phases[] = {
        {"[Receiver]\tEncrypting query vector... ",
         [&] { query = roles.receiver->encryptQuery(data.query); return std::to_string(query.size()); }},
        {"[Sender]\tComputing membership scenario... ",
         [&] { membership = roles.sender->membershipScenario(query); hydia_sync(cc->h); return std::string("1"); }},
        {"[Receiver]\tDecrypting membership results... ",
         [&] { is_member = roles.receiver->decryptMembership(membership); return std::string(); }},
        {"[Sender]\tComputing index scenario... ",
         [&] { index = roles.sender->indexScenario(query); hydia_sync(cc->h); return std::to_string(index.size()); }},
        {"[Receiver]\tDecrypting index results... ",
         [&] { hits = roles.receiver->decryptIndex(index); return std::string(); }},
    };
    std::cout << std::endl << "\tRunning Experiments:" << std::endl;
    for (const Phase &ph : phases) {
        std::cout << ph.label << std::flush;
        const auto t0 = std::chrono::steady_clock::now();
        const std::string cell = ph.run();
        const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::cout << "done (" << seconds << "s)" << std::endl;
        csv << seconds << "," << std::flush;
        if (!cell.empty()) csv << cell << "," << std::flush;
    }

    std::cout << std::endl << "\tDisplaying Query Results:" << std::endl;
    const char *verdict = is_member ? "true" : "false";
    std::cout << "Membership scenario: " << verdict << std::endl;
    std::cout << "Index scenario: " << show(hits) << std::endl;
    csv << verdict << "," << show(hits) << "," << std::endl;
    csv.close();
    std::cout << std::endl << "\tProgram successfully terminated" << std::endl;
    return cc->last_status == 0 ? 0 : 3;
}
