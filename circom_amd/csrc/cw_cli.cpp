// cw_witness — process-level drop-in for the reference's emitted calculator (SURVEY §8b seam 4):
//     reference:   ./<name> <input.json> <output.wtns>                 (main.cpp:336-373, <argv0>.dat beside it)
//     here:        cw_witness <name> <input.json> <output.wtns>        one instance, same files, same .wtns bytes
//                  cw_witness <name> <inputs.json> <out_%d.wtns>       inputs.json = JSON ARRAY of input objects:
//                                                                      one batch on the GPU, one .wtns per instance
// <name> is the path prefix of <name>.cwt / <name>.dat / <name>.r1cs (the .r1cs is optional: without it the
// constraint check is skipped).  Exit code 0 = every instance produced a witness; 1 = at least one failed (the
// reference aborts on the first failed assert, calcwit/assert_bucket.rs:75-77; here the others are still written).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/circom_amd.h"

static std::string slurp(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    std::string s;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
    fclose(f);
    return s;
}

// split a top-level JSON array into the texts of its elements (objects); a top-level object is one element
static std::vector<std::string> elements(const std::string &t) {
    size_t i = 0;
    while (i < t.size() && isspace((unsigned char)t[i])) i++;
    if (i >= t.size() || t[i] != '[') return {t};
    std::vector<std::string> out;
    int depth = 0;
    bool in_str = false;
    size_t start = 0;
    for (; i < t.size(); i++) {
        char ch = t[i];
        if (in_str) {
            if (ch == '\\') i++;
            else if (ch == '"') in_str = false;
            continue;
        }
        if (ch == '"') in_str = true;
        else if (ch == '{' || ch == '[') {
            if (depth == 1 && ch == '{') start = i;
            depth++;
        } else if (ch == '}' || ch == ']') {
            depth--;
            if (depth == 1 && ch == '}') out.push_back(t.substr(start, i - start + 1));
            if (depth == 0) break;
        }
    }
    return out;
}

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ != CW_OK) { fprintf(stderr, "%s: %s\n", #call, cw_last_error()); return 2; } \
    } while (0)

int main(int argc, char **argv) {
    if (argc != 4) {
        fprintf(stderr, "Usage: %s <circuit path prefix> <input.json> <output.wtns | out_%%d.wtns>\n", argv[0]);
        return 2;
    }
    const std::string name = argv[1];
    const std::string r1cs = name + ".r1cs";
    FILE *probe = fopen(r1cs.c_str(), "rb");
    if (probe) fclose(probe);
    cw_circuit *c = nullptr;
    CHECK(cw_load((name + ".cwt").c_str(), (name + ".dat").c_str(), probe ? r1cs.c_str() : nullptr, &c));
    // <name>.w2s beside the tape (written by `python -m circom_amd.circom` at its default level --O1): the signals the
    // simplified constraint system keeps, u32 little endian.  The files written below then hold THAT witness - what the
    // reference's calculator writes for the `.r1cs` it produced with the same flags.
    if (FILE *wl = fopen((name + ".w2s").c_str(), "rb")) {
        std::vector<uint32_t> list;
        uint32_t v;
        while (fread(&v, 4, 1, wl) == 1) list.push_back(v);
        fclose(wl);
        if (!list.empty()) CHECK(cw_set_witness_list(c, list.data(), (uint32_t)list.size()));
    }
    const std::vector<std::string> ins = elements(slurp(argv[2]));
    if (ins.empty()) { fprintf(stderr, "no input objects in %s\n", argv[2]); return 2; }
    const int device = getenv("CW_DEVICE") ? atoi(getenv("CW_DEVICE")) : 0;
    cw_batch *b = nullptr;
    CHECK(cw_batch_create(c, device, (uint32_t)ins.size(), nullptr, &b));
    for (size_t i = 0; i < ins.size(); i++) CHECK(cw_set_inputs_json(b, (uint32_t)i, ins[i].c_str()));
    CHECK(cw_run(b));
    if (probe) CHECK(cw_check_r1cs(b));
    CHECK(cw_sync(b));
    std::vector<uint32_t> st(ins.size());
    CHECK(cw_get_status(b, st.data()));
    int failed = 0;
    const bool many = strstr(argv[3], "%d") != nullptr;
    if (!many && ins.size() > 1) { fprintf(stderr, "%zu inputs need an output pattern with %%d\n", ins.size()); return 2; }
    const bool logs = cw_n_log_statements(c) != 0;
    for (size_t i = 0; i < ins.size(); i++) {
        if (logs) {                 // what the reference binary prints for this input (log(...) statements), instance by instance
            const int64_t n = cw_get_log(b, (uint32_t)i, nullptr, 0);
            if (n < 0) { fprintf(stderr, "cw_get_log: %s\n", cw_last_error()); return 2; }
            std::vector<char> text((size_t)n + 1);
            if (cw_get_log(b, (uint32_t)i, text.data(), text.size()) < 0) { fprintf(stderr, "cw_get_log: %s\n", cw_last_error()); return 2; }
            fwrite(text.data(), 1, (size_t)n, stdout);
        }
        if (st[i]) {
            fprintf(stderr, "instance %zu: %s%s%s\n", i, (st[i] & 1) ? "Failed assert " : "", (st[i] & 2) ? "division by zero " : "",
                    (st[i] & 4) ? "R1CS row violated" : "");
            failed++;
            continue;
        }
        char path[4096];
        if (many) snprintf(path, sizeof path, argv[3], (int)i);
        else snprintf(path, sizeof path, "%s", argv[3]);
        CHECK(cw_write_wtns(b, (uint32_t)i, path));
    }
    cw_batch_free(b);
    cw_free(c);
    return failed ? 1 : 0;
}
