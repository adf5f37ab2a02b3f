// dvt_prover_host — the reference's CLI verbs (src/main.rs:58-106) over the C ABI.
//
//   dvt_prover_host prove   --type T -i INPUT.json [-o PROOF] [--elf GUEST.elf]
//   dvt_prover_host execute --type T -i INPUT.json [--show-report] [--elf GUEST.elf]
//   dvt_prover_host verify  --type T -i PROOF [--elf GUEST.elf]
//
// T = bad-share | finalization | bad-partial-key | bad-encrypted-share (clap names of CircuitType, :36-42).
// The reference embeds the four guest ELFs at build time (include_elf!, :115-118); they cannot be built in
// this image (no RISC-V toolchain), so the ELF comes from --elf or from $DVT_ELF_DIR/<type>.elf.
// Same contract as the reference: default proof path "<input>_proof.bin" (:468-470), "Proof saved to:" on
// success (:476), any error is printed and the process exits with code 1 (:421-427) — which is what the
// reference's 92 test vectors observe (script/run.sh:82-89).  `verify` has stock client.verify semantics
// (the reference's own verify sub-command re-executes the guest instead).
// get-schema / validate-schema / node are product UI outside the accelerated path and are not provided.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../../include/dvt_prover.h"

static bool read_file(const std::string &path, std::vector<uint8_t> *out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    out->assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return true;
}
static int die(const std::string &m) {
    fprintf(stderr, "Error: %s\n", m.c_str());
    return 1;
}

int main(int argc, char **argv) {
    if (argc < 2) return die("usage: dvt_prover_host prove|execute|verify --type T -i FILE [-o FILE] [--show-report] [--elf FILE]");
    const std::string verb = argv[1];
    std::string type, input, output, elf_path, schema_path;
    bool show_report = false, auth = false;
    for (int i = 2; i < argc; i++) {
        std::string a = argv[i], inline_val;
        bool has_inline = false;
        if (a.rfind("--", 0) == 0) {   // --key=value, as the reference's test vectors write it (clap accepts both forms)
            size_t eq = a.find('=');
            if (eq != std::string::npos) { inline_val = a.substr(eq + 1); a = a.substr(0, eq); has_inline = true; }
        }
        auto next = [&]() -> std::string { return has_inline ? inline_val : (i + 1 < argc ? std::string(argv[++i]) : std::string()); };
        if (a == "--type") type = next();
        else if (a == "-i" || a == "--input-file") input = next();
        else if (a == "-o" || a == "--output-file-path") output = next();
        else if (a == "--elf") elf_path = next();
        else if (a == "--show-report") show_report = true;
        else if (a == "--auth-commitment") auth = true;  // the reference selects this at build time (cargo feature)
        else if (a == "--json-schema-file") schema_path = next();
        else return die("unknown argument " + a);
    }
    if (verb != "prove" && verb != "execute" && verb != "verify") return die("unknown sub-command " + verb);
    if (type.empty() || input.empty()) return die("--type and --input-file are required");
    if (elf_path.empty()) {
        const char *dir = getenv("DVT_ELF_DIR");
        if (!dir) return die("no guest ELF: pass --elf or set DVT_ELF_DIR (the reference embeds its guests at build time)");
        elf_path = std::string(dir) + "/" + type + ".elf";
    }
    std::vector<uint8_t> elf, in;
    if (!read_file(elf_path, &elf)) return die("cannot read ELF " + elf_path);
    if (!read_file(input, &in)) return die("cannot read " + input);

    if (!schema_path.empty() && verb != "verify") {  // validate_if_needed (src/main.rs:509-541)
        std::vector<uint8_t> sch;
        if (!read_file(schema_path, &sch)) return die("Could not read schema file '" + schema_path + "'");
        char *verr = nullptr;
        if (dvt_json_schema_validate((const char *)sch.data(), sch.size(), (const char *)in.data(), in.size(), &verr)) {
            std::string all = verr ? verr : "?";
            size_t at = 0;
            while (at <= all.size()) {
                size_t e = all.find('\n', at);
                fprintf(stderr, "Validation error in '%s': %s\n", input.c_str(), all.substr(at, e == std::string::npos ? std::string::npos : e - at).c_str());
                if (e == std::string::npos) break;
                at = e + 1;
            }
            return die("JSON validation failed for '" + input + "'");
        }
    }

    if (verb == "verify") {
        dvt_prover *p = nullptr;
        if (dvt_prover_create(nullptr, &p)) return die(dvt_last_error(nullptr));
        dvt_pk *pk = nullptr;
        uint8_t *vk = nullptr;
        size_t vk_len = 0;
        if (dvt_setup(p, elf.data(), elf.size(), &pk, &vk, &vk_len)) return die(dvt_last_error(p));
        char *why = nullptr;
        int32_t ec = 0;
        int rc = dvt_verify(vk, vk_len, in.data(), in.size(), 100, 16, &ec, nullptr, nullptr, &why);
        if (rc) return die(std::string("Verification failed: ") + (why ? why : "?"));
        printf("Proof verified (guest exit code %d)\n", ec);
        return 0;
    }

    uint8_t *stdin_buf = nullptr;
    size_t stdin_len = 0;
    char *err = nullptr;
    if (dvt_stdin_from_json(type.c_str(), (const char *)in.data(), in.size(), auth, &stdin_buf, &stdin_len, &err))
        return die(std::string("Failed to read input: ") + (err ? err : "?"));
    dvt_buf buf{stdin_buf, stdin_len};

    if (verb == "execute") {
        printf("input len: %zu\n", stdin_len - 8);  // the reference prints the CBOR length (src/main.rs:436)
        dvt_report rep{};
        int rc = dvt_execute(elf.data(), elf.size(), &buf, 1, 0, nullptr, nullptr, &rep, &err);
        if (rc) return die(std::string("Verification failed: ") + (err ? err : "?"));
        if (show_report) printf("Verification report:\ntotal instructions: %llu\nexit code: %d\n", (unsigned long long)rep.cycles, rep.exit_code);
        return 0;
    }

    dvt_prover *p = nullptr;
    if (dvt_prover_create(nullptr, &p)) return die(dvt_last_error(nullptr));
    dvt_pk *pk = nullptr;
    if (dvt_setup(p, elf.data(), elf.size(), &pk, nullptr, nullptr)) return die(dvt_last_error(p));
    uint8_t *proof = nullptr;
    size_t proof_len = 0;
    dvt_report rep{};
    if (dvt_prove_core(p, pk, &buf, 1, &proof, &proof_len, &rep)) return die(std::string("Proof generation failed: ") + dvt_last_error(p));
    const std::string path = output.empty() ? input + "_proof.bin" : output;
    std::ofstream f(path, std::ios::binary);
    if (!f || !f.write((const char *)proof, (std::streamsize)proof_len)) return die("Saving proof failed: " + path);
    printf("Proof saved to: %s\n", path.c_str());
    dvt_free(proof);
    dvt_pk_free(p, pk);
    dvt_prover_destroy(p);
    return 0;
}
