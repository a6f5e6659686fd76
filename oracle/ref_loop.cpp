// TEST INFRASTRUCTURE (oracle): batch driver around the *reference's* C++ runtime.
// The reference runs one process per input (main.cpp:336-373).  To (a) produce many .wtns files for
// parity checks and (b) time the compute-only path fairly (in-process loop over
// pre-parsed inputs calling run(ctx)), this driver links the reference objects unchanged
// (main.o is compiled with -Dmain=circom_ref_main so its loadCircuit/writeBinWitness are reusable,
// calcwit.o, fr.o, <circuit>.o) and loops over instances.
//
//   ref_loop <circuit.dat> <inputs.bin> <names.txt> <n> <reps> [wtns_prefix [stride]]
//     inputs.bin : [n][n_inputs][32] canonical little-endian values, main-input slot order
//     names.txt  : one "<name> <size>" line per main input, slot order
//     prints one JSON line: {"n":..,"reps":..,"seconds":..,"witnesses_per_s":..}
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>
#include <gmp.h>

#include "calcwit.hpp"
#include "circom.hpp"

Circom_Circuit *loadCircuit(std::string const &datFileName);          // main.cpp:22
void writeBinWitness(Circom_CalcWit *ctx, std::string wtnsFileName);   // main.cpp:288

// What Fr_str2element produces for a canonical value (Fr_fromMpz, generic/fr.cpp:2779-2788):
// short iff it fits a signed int, else long normal.
static void fe_from_le(FrElement *e, const uint8_t le[32]) {
    uint64_t v[4];
    memcpy(v, le, 32);
    if (v[1] == 0 && v[2] == 0 && v[3] == 0 && v[0] <= 0x7FFFFFFFULL) {
        e->type = Fr_SHORT;
        e->shortVal = (int32_t)v[0];
    } else {
        e->type = Fr_LONG;
        e->shortVal = 0;
        memcpy(e->longVal, v, 32);
    }
}

int main(int argc, char **argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: %s <dat> <inputs.bin> <names.txt> <n> <reps> [wtns_prefix [stride]]\n", argv[0]);
        return 2;
    }
    std::string dat = argv[1];
    long n = atol(argv[4]), reps = atol(argv[5]);
    std::string prefix = argc > 6 ? argv[6] : "";
    long stride = argc > 7 ? atol(argv[7]) : 1;
    std::vector<std::pair<u64, uint>> names;
    {
        std::ifstream f(argv[3]);
        std::string nm;
        uint sz;
        while (f >> nm >> sz) names.push_back({fnv1a(nm), sz});
    }
    uint n_in = get_main_input_signal_no();
    std::vector<uint8_t> in((size_t)n * n_in * 32);
    {
        FILE *f = fopen(argv[2], "rb");
        if (!f || fread(in.data(), 1, in.size(), f) != in.size()) { fprintf(stderr, "cannot read inputs\n"); return 2; }
        fclose(f);
    }
    Circom_Circuit *circuit = loadCircuit(dat);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (long r = 0; r < reps; r++) {
        for (long i = 0; i < n; i++) {
            Circom_CalcWit *ctx = new Circom_CalcWit(circuit);
            const uint8_t *p = &in[(size_t)i * n_in * 32];
            if (n_in == 0) ctx->tryRunCircuit();
            for (auto &nm : names) {
                for (uint k = 0; k < nm.second; k++) {
                    FrElement v;
                    fe_from_le(&v, p);
                    p += 32;
                    ctx->setInputSignal(nm.first, k, v);   // the last one triggers run(ctx) (calcwit.cpp:71-97)
                }
            }
            if (ctx->getRemaingInputsToBeSet() != 0) { fprintf(stderr, "Not all inputs have been set\n"); return 3; }
            if (!prefix.empty() && r == 0 && i % stride == 0) writeBinWitness(ctx, prefix + std::to_string(i) + ".wtns");
            // the reference never frees a context (one per process); free the big arrays here
            delete[] ctx->signalValues;
            delete[] ctx->componentMemory;
            delete ctx;
        }
    }
    double s = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    printf("{\"n\": %ld, \"reps\": %ld, \"seconds\": %.6f, \"witnesses_per_s\": %.3f}\n", n, reps, s, (double)n * reps / s);
    return 0;
}
