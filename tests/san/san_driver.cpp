// TEST-ONLY: drives the host side of libwfdwfst (waterfall_amd/csrc/wf_graph.cpp: prepare, topology index, numerator
// compiler, composers) under AddressSanitizer + UndefinedBehaviorSanitizer.  Built by tests/test_sanitizers.py with
//   g++ -fsanitize=address,undefined -fno-sanitize-recover=all -static-libasan -static-libubsan san_driver.cpp wf_graph.cpp
// (a CPU build: the HIP file is not part of it).  Input: a stream of int32 records written by the test
// ([opcode, then arrays as (length, data...)]); output: the results of every call as arrays in the same encoding, which
// the test compares with what the production library returns for the same inputs.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/wf_dwfst.h"

namespace {
std::vector<int32_t> in;
size_t pos = 0;
FILE* fout = nullptr;

std::vector<int32_t> arr() {
    if (pos >= in.size()) { fprintf(stderr, "truncated input\n"); exit(3); }
    const size_t n = (size_t)in[pos++];
    if (pos + n > in.size()) { fprintf(stderr, "truncated array\n"); exit(3); }
    // an exact-size heap copy per array: reads one element past any input land in an ASan redzone
    std::vector<int32_t> v(in.begin() + pos, in.begin() + pos + n);
    pos += n;
    return v;
}
int32_t scalar() { return in[pos++]; }
void put(const int32_t* p, size_t n) {
    const int32_t len = (int32_t)n;
    fwrite(&len, 4, 1, fout);
    if (n) fwrite(p, 4, n, fout);
}
void put1(int32_t v) { put(&v, 1); }
const int32_t* ptr(const std::vector<int32_t>& v) { return v.empty() ? nullptr : v.data(); }

void op_prepare() {
    auto arcs = arr(), off = arr(), ns = arr();
    wf_graph_host g{(int32_t)ns.size(), ptr(arcs), off.data(), ns.data()};
    size_t bytes = 0;
    int rc = wf_graph_prepare_bytes(&g, &bytes);
    if (rc) { put1(rc); return; }
    std::vector<int32_t> blob(bytes / 4);
    // a blob that is too small must be refused, not overrun
    if (bytes >= 64) {
        std::vector<int32_t> small(8);
        const int rs = wf_graph_prepare(&g, small.data(), 32);
        if (rs == WF_OK) { fprintf(stderr, "prepare accepted a 32-byte blob\n"); exit(4); }
    }
    rc = wf_graph_prepare(&g, blob.data(), bytes);
    put1(rc);
    if (rc == WF_OK) put(blob.data(), (size_t)blob[2]);
}

void op_numerator() {
    auto arcs = arr(), aux = arr();
    const int32_t S = scalar(), ctc_cross = scalar();
    auto pstate = arr(), ptoken = arr(), phones = arr(), poff = arr();
    const int32_t nthreads = scalar();
    const int32_t nA = (int32_t)aux.size(), B = (int32_t)poff.size() - 1;
    int64_t words = 0;
    int rc = wf_topo_index_words(arcs.data(), aux.data(), nA, S, &words);
    if (rc) { put1(rc); return; }
    std::vector<int32_t> index((size_t)words);
    rc = wf_topo_index_build(arcs.data(), aux.data(), nA, S, index.data());
    if (rc) { put1(rc); return; }
    std::vector<int32_t> arc_off(B + 1), nst(B);
    std::vector<int32_t> ph = phones.empty() ? std::vector<int32_t>(1, 0) : phones;
    // first with no room at all: the sizes must come back with WF_ENOSPC
    size_t need = 0;
    std::vector<int32_t> oa(4), ox(1), tiny(16);
    rc = wf_numerator_compile(arcs.data(), aux.data(), nA, S, index.data(), ctc_cross, ptr(pstate), ptr(ptoken),
                              (int32_t)pstate.size(), ph.data(), poff.data(), B, oa.data(), ox.data(), 1, arc_off.data(),
                              nst.data(), tiny.data(), tiny.size() * 4, &need, nthreads);
    if (rc != WF_ENOSPC) { put1(rc == WF_OK ? -100 : rc); return; }
    const size_t cap = (size_t)arc_off[B];
    oa.assign(4 * cap, 0); ox.assign(cap, 0);
    std::vector<int32_t> blob(need / 4);
    rc = wf_numerator_compile(arcs.data(), aux.data(), nA, S, index.data(), ctc_cross, ptr(pstate), ptr(ptoken),
                              (int32_t)pstate.size(), ph.data(), poff.data(), B, oa.data(), ox.data(), (int64_t)cap,
                              arc_off.data(), nst.data(), blob.data(), need, &need, nthreads);
    put1(rc);
    if (rc) return;
    put(oa.data(), 4 * (size_t)arc_off[B]); put(ox.data(), (size_t)arc_off[B]);
    put(arc_off.data(), arc_off.size()); put(nst.data(), nst.size());
    // the graphs sit at the start of upper-bound regions: dump every graph's info words and the header, not the gaps
    put(blob.data(), 8 + 64 * (size_t)B);
    // the same batch through the two-step route (compose, then prepare): must give the same FsaVec
    std::vector<int32_t> oa2(4 * cap), ox2(cap), off2(B + 1), ns2(B);
    rc = wf_compose_linear_batch(arcs.data(), aux.data(), nA, S, index.data(), ctc_cross, ptr(pstate), ptr(ptoken),
                                 (int32_t)pstate.size(), ph.data(), poff.data(), B, oa2.data(), ox2.data(), (int64_t)cap,
                                 off2.data(), ns2.data());
    put1(rc);
    put1(rc == WF_OK && off2 == arc_off && ns2 == nst && !memcmp(oa2.data(), oa.data(), 16 * cap) && !memcmp(ox2.data(), ox.data(), 4 * cap));
}

void op_compose_fsa() {
    auto arcs = arr(), aux = arr();
    const int32_t S = scalar();
    auto barcs = arr(), baux = arr(), boff = arr(), bns = arr();
    const int32_t nA = (int32_t)aux.size(), B = (int32_t)bns.size();
    int64_t words = 0;
    int rc = wf_topo_index_words(arcs.data(), aux.data(), nA, S, &words);
    if (rc) { put1(rc); return; }
    std::vector<int32_t> index((size_t)words);
    rc = wf_topo_index_build(arcs.data(), aux.data(), nA, S, index.data());
    if (rc) { put1(rc); return; }
    std::vector<int32_t> arc_off(B + 1), nst(B > 0 ? B : 1);
    rc = wf_compose_fsa_batch(arcs.data(), aux.data(), nA, S, index.data(), barcs.data(), ptr(baux), boff.data(), bns.data(), B,
                              nullptr, nullptr, 0, arc_off.data(), nst.data());
    if (rc != WF_OK && rc != WF_ENOSPC) { put1(rc); return; }
    const size_t cap = (size_t)arc_off[B];
    std::vector<int32_t> oa(4 * cap + 4), ox(cap + 1);
    rc = wf_compose_fsa_batch(arcs.data(), aux.data(), nA, S, index.data(), barcs.data(), ptr(baux), boff.data(), bns.data(), B,
                              oa.data(), ox.data(), (int64_t)cap, arc_off.data(), nst.data());
    put1(rc);
    if (rc) return;
    put(oa.data(), 4 * cap); put(ox.data(), cap); put(arc_off.data(), arc_off.size()); put(nst.data(), (size_t)B);
}

void op_compose_one() {
    auto arcs = arr(), aux = arr();
    const int32_t S = scalar();
    auto phones = arr();
    int32_t na = 0, ns = 0;
    int rc = wf_compose_linear(arcs.data(), aux.data(), (int32_t)aux.size(), S, ptr(phones), (int32_t)phones.size(), nullptr,
                               nullptr, &na, &ns, 0);
    if (rc) { put1(rc); return; }
    std::vector<int32_t> oa(4 * (size_t)na), ox((size_t)na);
    rc = wf_compose_linear(arcs.data(), aux.data(), (int32_t)aux.size(), S, ptr(phones), (int32_t)phones.size(), oa.data(),
                           ox.data(), &na, &ns, na);
    put1(rc);
    if (rc) return;
    put(oa.data(), oa.size()); put(ox.data(), ox.size()); put1(ns);
}
}  // namespace

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: san_driver <input> <output>\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    in.resize((size_t)sz / 4);
    if (fread(in.data(), 4, in.size(), f) != in.size()) return 2;
    fclose(f);
    fout = fopen(argv[2], "wb");
    if (!fout) return 2;
    while (pos < in.size()) {
        const int32_t op = scalar();
        if (op == 1) op_prepare();
        else if (op == 2) op_numerator();
        else if (op == 3) op_compose_fsa();
        else if (op == 4) op_compose_one();
        else { fprintf(stderr, "unknown opcode %d\n", op); return 3; }
    }
    fclose(fout);
    return 0;
}
