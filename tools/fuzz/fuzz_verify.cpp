// Byte-mutation fuzz driver of the UNTRUSTED-input side of the library: dvt_verify (container parsing in csrc/capi.hip,
// csrc/proof.h, the shard verifier csrc/verifier.hip).  Built host-only with AddressSanitizer + UBSan by
// `make -C dvt_circuits_amd/csrc asan-fuzz` (no GPU involved: sanitizers run on the CPU build only) and run by
// tests/test_verify_fuzz.py on a proof fixture made on a GPU box (tools/make_proof_fixture.py).
//
//   fuzz_verify <fixture> <iterations> <seed> <fri_queries> <pow_bits>
// fixture = u32 vk_len | vk | proof.  Every mutated proof must come back as a clean DVT_ERR_REJECTED / DVT_ERR_INPUT
// (or DVT_OK when the mutation left the proof valid, which only the pristine copy does); the sanitizers abort on any
// out-of-bounds read, overflow or other undefined behaviour on the way.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dvt_prover.h"

static uint64_t rng_state;
static uint64_t rnd() {   // xorshift64*
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return rng_state * 0x2545F4914F6CDD1Dull;
}

int main(int argc, char **argv) {
    if (argc < 6) { fprintf(stderr, "usage: fuzz_verify fixture iterations seed fri_queries pow_bits\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("fixture"); return 2; }
    std::vector<uint8_t> all;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) all.insert(all.end(), buf, buf + n);
    fclose(f);
    if (all.size() < 8) return 2;
    uint32_t vk_len;
    memcpy(&vk_len, all.data(), 4);
    if (4 + (size_t)vk_len >= all.size()) return 2;
    const std::vector<uint8_t> vk(all.begin() + 4, all.begin() + 4 + vk_len), proof(all.begin() + 4 + vk_len, all.end());
    const long iters = atol(argv[2]);
    rng_state = strtoull(argv[3], nullptr, 10) * 2 + 1;
    const uint32_t q = (uint32_t)atoi(argv[4]), pw = (uint32_t)atoi(argv[5]);
    auto run = [&](const std::vector<uint8_t> &v, const std::vector<uint8_t> &p) {
        int32_t ec = 0; uint8_t *pv = nullptr; size_t pvl = 0; char *why = nullptr;
        int rc = dvt_verify(v.data(), v.size(), p.data(), p.size(), q, pw, &ec, &pv, &pvl, &why);
        if (pv) dvt_free(pv);
        if (why) dvt_free(why);
        return rc;
    };
    if (run(vk, proof) != DVT_OK) { fprintf(stderr, "the pristine fixture does not verify\n"); return 3; }
    long ok = 0, rejected = 0, input = 0, other = 0;
    for (long it = 0; it < iters; it++) {
        std::vector<uint8_t> p = proof, v = vk;
        const int kind = (int)(rnd() % 8);
        if (kind == 0) {                                  // truncate
            p.resize(rnd() % p.size());
        } else if (kind == 1) {                           // a length / count word becomes huge or tiny
            size_t at = (rnd() % (p.size() / 4)) * 4;
            uint32_t w = (rnd() & 1) ? 0xFFFFFFFFu : (uint32_t)(rnd() % 5);
            memcpy(&p[at], &w, 4);
        } else if (kind == 2) {                           // mutate the verifying key instead
            v[rnd() % v.size()] ^= (uint8_t)(1u << (rnd() % 8));
        } else if (kind == 3) {                           // splice: copy a random window over another place
            size_t len = 1 + rnd() % 64, a = rnd() % (p.size() - len), b = rnd() % (p.size() - len);
            memmove(&p[a], &p[b], len);
        } else if (kind == 4) {                           // append garbage
            for (int k = 0; k < 8; k++) p.push_back((uint8_t)rnd());
        } else {                                          // 1..4 random byte changes (biased to the structured head of the container)
            int m = 1 + (int)(rnd() % 4);
            for (int k = 0; k < m; k++) {
                size_t at = (rnd() & 3) ? rnd() % p.size() : rnd() % (p.size() < 4096 ? p.size() : 4096);
                p[at] ^= (uint8_t)(1 + rnd() % 255);
            }
        }
        const int rc = run(v, p);
        if (rc == DVT_OK && getenv("FUZZ_VERBOSE")) {
            size_t first = 0;
            while (first < p.size() && first < proof.size() && p[first] == proof[first]) first++;
            fprintf(stderr, "still valid: kind %d, first changed byte %zu of %zu (sizes %zu / %zu, vk changed %d)\n", kind, first, proof.size(), p.size(), proof.size(), (int)(v != vk));
        }
        if (rc == DVT_OK) ok++;
        else if (rc == DVT_ERR_REJECTED) rejected++;
        else if (rc == DVT_ERR_INPUT) input++;
        else other++;
    }
    printf("{\"iterations\": %ld, \"ok\": %ld, \"rejected\": %ld, \"input\": %ld, \"other\": %ld}\n", iters, ok, rejected, input, other);
    return other ? 4 : 0;
}
