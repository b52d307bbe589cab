// image_matching_amd/csrc/cli_main.cpp — `ImageMatching <file.dat> 5`: the reference's latency driver for approach 5
// (/root/reference/src/main.cpp:40-110 argument handling, :216-247 dataset + enrolment, :330-393 the five timed phases,
// latency.csv row with the columns of /root/reference/tools/setup_experiment.sh:4-16) on the MI355X stack.
// Approach 4 (HERS) runs on the same kernels; approaches 1-3 are not part of this framework.
#include <chrono>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../include/hydia_roles.hpp"

using namespace std;
using namespace hydia;

static const string EXP_FILEPATH = "latency.csv";  // include/config.h:36

static ostream &operator<<(ostream &os, const vector<size_t> &v) {  // OpenFHE prints vectors as "[ a b c ]"
    os << "[ ";
    for (size_t x : v) os << x << " ";
    return os << "]";
}

int main(int argc, char *argv[]) {
    cout << "\tRunning Setup Operations:" << endl;
    ifstream fileStream;
    if (argc > 1) {
        fileStream.open(argv[1], ios::in);
    } else {
        cerr << "Error: input file not included" << endl;
        return 1;
    }
    if (!fileStream.is_open()) {
        cerr << "Error: unable to open input file" << endl;
        return 1;
    }
    size_t numVectors;
    fileStream >> numVectors;
    size_t expApproach;
    if (argc > 2) {
        expApproach = atoi(argv[2]);
    } else {
        cerr << "Error: approach argument not included" << endl;
        return 1;
    }
    if (expApproach < 1 || expApproach > 5) {
        cerr << "Error: approach must be from 1 to 5" << endl;
        return 1;
    }
    if (expApproach != 5 && expApproach != 4) {
        cerr << "Error: only approach 5 (novel diagonal transform, HyDia) and approach 4 (HERS) are built in hydia-mi355x" << endl;
        return 1;
    }
    ofstream expStream;
    expStream.open(EXP_FILEPATH, ios::app);
    if (!expStream.is_open()) {
        cerr << "Error: experiment file not found" << endl;
        return 1;
    }
    size_t multDepth = OpenFHEWrapper::computeRequiredDepth(expApproach);
    if (expApproach == 5) {
        cout << "Experimental approach: Novel diagonal transform" << endl;
        expStream << "Diagonal," << flush;
    } else {
        cout << "Experimental approach: HERS paper" << endl;
        expStream << "HERS," << flush;
    }

    CryptoContext cc = GenCryptoContext(multDepth, 45, VECTOR_DIM);
    if (!cc->h) return 2;
    size_t batchSize = cc->GetBatchSize();
    cout << "Generating key pair, mult keys, sum keys and rotation keys on the GPU... " << endl;
    uint8_t seed[32];
    for (int i = 0; i < 32; i++) seed[i] = (uint8_t)(i * 7 + 1);
    if (!cc->KeyGen(seed)) return 2;
    cout << "CKKS scheme set up (depth = " << multDepth << ", batch size = " << batchSize << ")" << endl;
    expStream << numVectors << "," << flush;

    vector<double> queryVector(VECTOR_DIM);
    for (size_t i = 0; i < VECTOR_DIM; i++) fileStream >> queryVector[i];
    cout << "Reading database vectors from file... " << endl;
    vector<vector<double>> plaintextVectors(numVectors, vector<double>(VECTOR_DIM));
    for (size_t i = 0; i < numVectors; i++)
        for (size_t j = 0; j < VECTOR_DIM; j++) fileStream >> plaintextVectors[i][j];
    fileStream.close();
    cout << "Encrypting database vectors... " << endl;
    if (expApproach == 5) {
        DiagonalEnroller enroller(cc, numVectors);
        enroller.serializeDB(plaintextVectors);
    } else {
        HersEnroller enroller(cc, numVectors);
        enroller.serializeDB(plaintextVectors);
    }

    cout << endl << "\tRunning Experiments:" << endl;
    chrono::steady_clock::time_point start, end;
    chrono::duration<double> duration;
    Receiver *receiver = expApproach == 5 ? (Receiver *)new DiagonalReceiver(cc, numVectors) : (Receiver *)new HersQueryReceiver(cc, numVectors);
    Sender *sender = expApproach == 5 ? (Sender *)new DiagonalSender(cc, numVectors) : (Sender *)new HersSender(cc, numVectors);

    cout << "[Receiver]\tEncrypting query vector... " << flush;
    start = chrono::steady_clock::now();
    vector<Ciphertext> queryCipher = receiver->encryptQuery(queryVector);
    end = chrono::steady_clock::now();
    duration = end - start;
    cout << "done (" << duration.count() << "s)" << endl;
    expStream << duration.count() << "," << queryCipher.size() << "," << flush;

    cout << "[Sender]\tComputing membership scenario... " << flush;
    start = chrono::steady_clock::now();
    Ciphertext membershipCipher = sender->membershipScenario(queryCipher);
    hydia_sync(cc->h);
    end = chrono::steady_clock::now();
    duration = end - start;
    cout << "done (" << duration.count() << "s)" << endl;
    expStream << duration.count() << "," << 1 << "," << flush;

    cout << "[Receiver]\tDecrypting membership results... " << flush;
    start = chrono::steady_clock::now();
    bool membershipResult = receiver->decryptMembership(membershipCipher);
    end = chrono::steady_clock::now();
    duration = end - start;
    cout << "done (" << duration.count() << "s)" << endl;
    expStream << duration.count() << "," << flush;

    cout << "[Sender]\tComputing index scenario... " << flush;
    start = chrono::steady_clock::now();
    auto indexCipher = sender->indexScenario(queryCipher);
    hydia_sync(cc->h);
    end = chrono::steady_clock::now();
    duration = end - start;
    cout << "done (" << duration.count() << "s)" << endl;
    expStream << duration.count() << "," << indexCipher.size() << "," << flush;

    cout << "[Receiver]\tDecrypting index results... " << flush;
    start = chrono::steady_clock::now();
    vector<size_t> indexResults = receiver->decryptIndex(indexCipher);
    end = chrono::steady_clock::now();
    duration = end - start;
    cout << "done (" << duration.count() << "s)" << endl;
    expStream << duration.count() << "," << flush;

    cout << endl << "\tDisplaying Query Results:" << endl;
    cout << "Membership scenario: " << (membershipResult ? "true" : "false") << endl;
    expStream << (membershipResult ? "true" : "false") << "," << flush;
    cout << "Index scenario: " << indexResults << endl;
    expStream << indexResults << "," << flush;
    expStream << endl;
    expStream.close();
    delete receiver;
    delete sender;
    cout << endl << "\tProgram successfully terminated" << endl;
    return cc->last_status == 0 ? 0 : 3;
}
