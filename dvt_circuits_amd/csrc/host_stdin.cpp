// Host side of the reference's prove()/execute() (src/main.rs:430-478): typed JSON input ->
// CBOR (`serde_cbor::to_vec(data)`, :435,:459) -> the single SP1Stdin buffer the guest reads
// (`stdin.write(&bin)`, :437,:460 = bincode of a Vec<u8>: u64-LE length then the bytes).
//
// The input types are the reference's (crates/dkg/src/types.rs): structs serialise as CBOR maps
// with their serde names in declaration order (:26-203), fixed-size byte arrays as lowercase hex
// TEXT strings (:322-330), u8 as unsigned ints, Vec as definite-length arrays.  Unknown JSON keys
// are ignored (no deny_unknown_fields); a missing or malformed field is an input error, as in
// the reference's typed deserialisation (`src/file_utils.rs:24-32`).  The host is instantiated with
// BLS keys and secp256k1 identity keys (src/main.rs:421): sizes :396-407.
// PARITY UNPINNED: serde_cbor is not in the container; the byte layout follows its published
// behaviour (RFC 8949 definite-length major types) and is cross-checked against an independent
// Python encoder in tests/test_host_stdin.py.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <cmath>
#include <memory>
#include <regex>
#include <string>
#include <vector>

#include "../../include/dvt_prover.h"

namespace {

// ------------------------------------------------------------------ minimal JSON
struct JVal {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal *get(const std::string &k) const {
        for (auto &kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
};

struct JParser {
    const char *p, *end;
    std::string err;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    bool fail(const char *m) { if (err.empty()) err = m; return false; }
    bool parse_string(std::string *out) {
        if (p >= end || *p != '"') return fail("expected string");
        p++;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return fail("bad escape");
                switch (*p) {
                case 'n': out->push_back('\n'); break;
                case 't': out->push_back('\t'); break;
                case 'r': out->push_back('\r'); break;
                case 'b': out->push_back('\b'); break;
                case 'f': out->push_back('\f'); break;
                case 'u': {
                    if (end - p < 5) return fail("bad \\u escape");
                    unsigned cp = (unsigned)strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
                    p += 4;
                    if (cp < 0x80) out->push_back((char)cp);
                    else if (cp < 0x800) { out->push_back((char)(0xC0 | (cp >> 6))); out->push_back((char)(0x80 | (cp & 0x3F))); }
                    else { out->push_back((char)(0xE0 | (cp >> 12))); out->push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out->push_back((char)(0x80 | (cp & 0x3F))); }
                    break;
                }
                default: out->push_back(*p);
                }
                p++;
            } else out->push_back(*p++);
        }
        if (p >= end) return fail("unterminated string");
        p++;
        return true;
    }
    bool parse(JVal *v, int depth = 0) {
        if (depth > 64) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end");
        if (*p == '{') {
            v->kind = JVal::Obj;
            p++; ws();
            if (p < end && *p == '}') { p++; return true; }
            for (;;) {
                ws();
                std::string k;
                if (!parse_string(&k)) return false;
                ws();
                if (p >= end || *p != ':') return fail("expected ':'");
                p++;
                JVal c;
                if (!parse(&c, depth + 1)) return false;
                v->obj.emplace_back(std::move(k), std::move(c));
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == '}') { p++; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (*p == '[') {
            v->kind = JVal::Arr;
            p++; ws();
            if (p < end && *p == ']') { p++; return true; }
            for (;;) {
                JVal c;
                if (!parse(&c, depth + 1)) return false;
                v->arr.push_back(std::move(c));
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == ']') { p++; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (*p == '"') { v->kind = JVal::Str; return parse_string(&v->str); }
        if (!strncmp(p, "true", 4) && end - p >= 4) { v->kind = JVal::Bool; v->b = true; p += 4; return true; }
        if (!strncmp(p, "false", 5) && end - p >= 5) { v->kind = JVal::Bool; p += 5; return true; }
        if (!strncmp(p, "null", 4) && end - p >= 4) { p += 4; return true; }
        char *e = nullptr;
        v->num = strtod(p, &e);
        if (e == p) return fail("unexpected character");
        v->kind = JVal::Num;
        p = e;
        return true;
    }
};

// ------------------------------------------------------------------ CBOR (RFC 8949, definite lengths)
struct Cbor {
    std::vector<uint8_t> out;
    void head(uint8_t major, uint64_t n) {
        uint8_t m = (uint8_t)(major << 5);
        if (n < 24) out.push_back(m | (uint8_t)n);
        else if (n < 0x100) { out.push_back(m | 24); out.push_back((uint8_t)n); }
        else if (n < 0x10000) { out.push_back(m | 25); out.push_back((uint8_t)(n >> 8)); out.push_back((uint8_t)n); }
        else if (n < 0x100000000ull) { out.push_back(m | 26); for (int s = 24; s >= 0; s -= 8) out.push_back((uint8_t)(n >> s)); }
        else { out.push_back(m | 27); for (int s = 56; s >= 0; s -= 8) out.push_back((uint8_t)(n >> s)); }
    }
    void uint(uint64_t v) { head(0, v); }
    void text(const std::string &s) { head(3, s.size()); out.insert(out.end(), s.begin(), s.end()); }
    void array(size_t n) { head(4, n); }
    void map(size_t n) { head(5, n); }
};

// ------------------------------------------------------------------ the reference's input schema
enum FieldKind { F_U8, F_HEX, F_STRING, F_STRUCT, F_VEC_HEX, F_VEC_STRUCT };
struct Schema;
struct Field {
    const char *name;
    FieldKind kind;
    size_t hex_bytes;      // F_HEX / F_VEC_HEX: byte length of the raw type
    const Schema *sub;     // F_STRUCT / F_VEC_STRUCT
    bool auth_only;        // present only with the reference's `auth_commitment` feature (types.rs:71-78)
};
struct Schema {
    const char *name;
    std::vector<Field> fields;
};

constexpr size_t BLS_PK = 48, BLS_SIG = 96, BLS_SK = 32, GEN_ID = 16, SHA = 32, SECP_PK = 33, SECP_SIG = 64;

const Schema S_SETTINGS = {"GenerateSettings", {{"n", F_U8, 0, nullptr, false}, {"k", F_U8, 0, nullptr, false}, {"gen_id", F_HEX, GEN_ID, nullptr, false}}};
const Schema S_INITIAL = {"InitialCommitment", {{"hash", F_HEX, SHA, nullptr, false}, {"settings", F_STRUCT, 0, &S_SETTINGS, false}, {"base_pubkeys", F_VEC_HEX, BLS_PK, nullptr, false}}};
const Schema S_SECRET = {"ExchangedSecret", {{"dst_base_hash", F_HEX, SHA, nullptr, false}, {"shared_secret", F_HEX, BLS_SK, nullptr, false}}};
const Schema S_COMMITMENT = {"Commitment", {{"hash", F_HEX, SHA, nullptr, true}, {"pubkey", F_HEX, SECP_PK, nullptr, false}, {"signature", F_HEX, SECP_SIG, nullptr, true}}};
const Schema S_SEED_EXCHANGE = {"SeedExchangeCommitment", {{"initial_commitment_hash", F_HEX, SHA, nullptr, false}, {"ssecret", F_STRUCT, 0, &S_SECRET, false}, {"commitment", F_STRUCT, 0, &S_COMMITMENT, false}}};
const Schema S_SHARED_DATA = {"SharedData", {{"base_hashes", F_VEC_HEX, SHA, nullptr, false}, {"initial_commitment", F_STRUCT, 0, &S_INITIAL, false}, {"seeds_exchange_commitment", F_STRUCT, 0, &S_SEED_EXCHANGE, false}}};
const Schema S_GENERATION = {"Generation", {{"base_pubkeys", F_VEC_HEX, BLS_PK, nullptr, false}, {"base_hash", F_HEX, SHA, nullptr, false}, {"partial_pubkey", F_HEX, BLS_PK, nullptr, false}, {"message_cleartext", F_STRING, 0, nullptr, false}, {"message_signature", F_HEX, BLS_SIG, nullptr, false}}};
const Schema S_FINALIZATION = {"FinalizationData", {{"settings", F_STRUCT, 0, &S_SETTINGS, false}, {"generations", F_VEC_STRUCT, 0, &S_GENERATION, false}, {"aggregate_pubkey", F_HEX, BLS_PK, nullptr, false}}};
const Schema S_BAD_GEN = {"BadPartialShareGeneration", {{"base_pubkeys", F_VEC_HEX, BLS_PK, nullptr, false}, {"base_hash", F_HEX, SHA, nullptr, false}}};
const Schema S_BAD_PARTIAL = {"BadPartialShare", {{"settings", F_STRUCT, 0, &S_SETTINGS, false}, {"data", F_STRUCT, 0, &S_GENERATION, false}, {"commitment", F_STRUCT, 0, &S_COMMITMENT, false}}};
const Schema S_BAD_PARTIAL_DATA = {"BadPartialShareData", {{"settings", F_STRUCT, 0, &S_SETTINGS, false}, {"generations", F_VEC_STRUCT, 0, &S_BAD_GEN, false}, {"bad_partial", F_STRUCT, 0, &S_BAD_PARTIAL, false}}};
const Schema S_BAD_ENCRYPTED = {"BadEncryptedShare", {{"sender_pubkey", F_HEX, SECP_PK, nullptr, false}, {"sender_encr_pubkey", F_HEX, BLS_PK, nullptr, false}, {"receiver_encr_seckey", F_HEX, BLS_SK, nullptr, false}, {"encrypted_data", F_STRING, 0, nullptr, false}, {"settings", F_STRUCT, 0, &S_SETTINGS, false}, {"base_hashes", F_VEC_HEX, SHA, nullptr, false}, {"sender_base_pubkeys", F_VEC_HEX, BLS_PK, nullptr, false}, {"receiver_base_pubkeys", F_VEC_HEX, BLS_PK, nullptr, false}}};

const Schema *schema_for(const std::string &type) {   // clap value names of CircuitType (src/main.rs:36-42)
    if (type == "bad-share") return &S_SHARED_DATA;
    if (type == "finalization") return &S_FINALIZATION;
    if (type == "bad-partial-key") return &S_BAD_PARTIAL_DATA;
    if (type == "bad-encrypted-share") return &S_BAD_ENCRYPTED;
    return nullptr;
}

bool hex_ok(const std::string &s, size_t bytes, std::string *lower) {
    if (s.size() != 2 * bytes) return false;
    lower->clear();
    for (char ch : s) {
        if (ch >= '0' && ch <= '9') lower->push_back(ch);
        else if (ch >= 'a' && ch <= 'f') lower->push_back(ch);
        else if (ch >= 'A' && ch <= 'F') lower->push_back((char)(ch - 'A' + 'a'));  // hex::decode accepts upper case, hex::encode emits lower
        else return false;
    }
    return true;
}

bool encode(const Schema &s, const JVal &v, bool auth, Cbor *c, std::string *err, const std::string &path) {
    if (v.kind != JVal::Obj) { *err = path + ": expected an object (" + s.name + ")"; return false; }
    size_t n = 0;
    for (auto &f : s.fields) if (!f.auth_only || auth) n++;
    c->map(n);
    for (auto &f : s.fields) {
        if (f.auth_only && !auth) continue;
        const JVal *x = v.get(f.name);
        const std::string here = path + "." + f.name;
        if (!x) { *err = "missing field `" + std::string(f.name) + "` at " + path; return false; }
        c->text(f.name);
        std::string lower;
        switch (f.kind) {
        case F_U8:
            if (x->kind != JVal::Num || x->num < 0 || x->num > 255 || x->num != (double)(uint64_t)x->num) { *err = here + ": expected u8"; return false; }
            c->uint((uint64_t)x->num);
            break;
        case F_HEX:
            if (x->kind != JVal::Str || !hex_ok(x->str, f.hex_bytes, &lower)) { *err = here + ": expected " + std::to_string(f.hex_bytes) + " bytes of hex"; return false; }
            c->text(lower);
            break;
        case F_STRING:
            if (x->kind != JVal::Str) { *err = here + ": expected a string"; return false; }
            c->text(x->str);
            break;
        case F_STRUCT:
            if (!encode(*f.sub, *x, auth, c, err, here)) return false;
            break;
        case F_VEC_HEX:
            if (x->kind != JVal::Arr) { *err = here + ": expected an array"; return false; }
            c->array(x->arr.size());
            for (size_t i = 0; i < x->arr.size(); i++) {
                if (x->arr[i].kind != JVal::Str || !hex_ok(x->arr[i].str, f.hex_bytes, &lower)) { *err = here + "[" + std::to_string(i) + "]: expected " + std::to_string(f.hex_bytes) + " bytes of hex"; return false; }
                c->text(lower);
            }
            break;
        case F_VEC_STRUCT:
            if (x->kind != JVal::Arr) { *err = here + ": expected an array"; return false; }
            c->array(x->arr.size());
            for (size_t i = 0; i < x->arr.size(); i++)
                if (!encode(*f.sub, x->arr[i], auth, c, err, here + "[" + std::to_string(i) + "]")) return false;
            break;
        }
    }
    return true;
}

// ------------------------------------------------------------------ JSON-schema check (draft-07 subset)
// What the reference does with --json-schema-file (src/main.rs:509-541: JSONSchema::compile + validate, every error
// printed, then "JSON validation failed").  Keywords: exactly those of the reference's own schemas (spec/json/*.json,
// generated from its Rust types): type, $ref into #/definitions, required, properties, items, minLength, maxLength,
// pattern, minimum / maximum; annotations ($schema, title, description, format, definitions) are ignored, and so is any
// other keyword, as the draft prescribes for unknown ones.
struct SchemaCheck {
    const JVal &root;
    std::vector<std::string> errors;
    const JVal *resolve(const JVal &sch, int depth) {
        const JVal *cur = &sch;
        while (cur && cur->kind == JVal::Obj && depth++ < 32) {
            const JVal *ref = cur->get("$ref");
            if (!ref || ref->kind != JVal::Str) return cur;
            const std::string &r = ref->str;
            if (r.rfind("#/", 0) != 0) { errors.push_back("unsupported $ref `" + r + "`"); return nullptr; }
            const JVal *t = &root;
            size_t at = 2;
            while (t && at <= r.size()) {
                size_t e = r.find('/', at);
                std::string key = r.substr(at, e == std::string::npos ? std::string::npos : e - at);
                t = t->kind == JVal::Obj ? t->get(key) : nullptr;
                if (e == std::string::npos) break;
                at = e + 1;
            }
            if (!t) { errors.push_back("unresolvable $ref `" + r + "`"); return nullptr; }
            cur = t;
        }
        return cur;
    }
    static bool is_type(const JVal &v, const std::string &t) {
        if (t == "object") return v.kind == JVal::Obj;
        if (t == "array") return v.kind == JVal::Arr;
        if (t == "string") return v.kind == JVal::Str;
        if (t == "boolean") return v.kind == JVal::Bool;
        if (t == "null") return v.kind == JVal::Null;
        if (t == "number") return v.kind == JVal::Num;
        if (t == "integer") return v.kind == JVal::Num && std::floor(v.num) == v.num;
        return true;
    }
    void check(const JVal &sch0, const JVal &v, const std::string &at) {
        const JVal *sch = resolve(sch0, 0);
        if (!sch || sch->kind != JVal::Obj) return;
        if (const JVal *t = sch->get("type")) {
            bool ok = false;
            if (t->kind == JVal::Str) ok = is_type(v, t->str);
            else if (t->kind == JVal::Arr) for (auto &x : t->arr) ok = ok || (x.kind == JVal::Str && is_type(v, x.str));
            else ok = true;
            if (!ok) { errors.push_back(at + ": is not of type " + (t->kind == JVal::Str ? "\"" + t->str + "\"" : "in the list")); return; }
        }
        if (v.kind == JVal::Str) {
            // (lengths count characters; the reference's strings are ASCII hex)
            if (const JVal *m = sch->get("minLength")) if (m->kind == JVal::Num && v.str.size() < (size_t)m->num) errors.push_back(at + ": is shorter than " + std::to_string((long)m->num) + " characters");
            if (const JVal *m = sch->get("maxLength")) if (m->kind == JVal::Num && v.str.size() > (size_t)m->num) errors.push_back(at + ": is longer than " + std::to_string((long)m->num) + " characters");
            if (const JVal *m = sch->get("pattern"))
                if (m->kind == JVal::Str) {
                    try {
                        if (!std::regex_search(v.str, std::regex(m->str, std::regex::ECMAScript))) errors.push_back(at + ": does not match \"" + m->str + "\"");
                    } catch (const std::regex_error &) { errors.push_back(at + ": schema pattern \"" + m->str + "\" is not a valid regular expression"); }
                }
        }
        if (v.kind == JVal::Num) {
            if (const JVal *m = sch->get("minimum")) if (m->kind == JVal::Num && v.num < m->num) errors.push_back(at + ": is less than the minimum");
            if (const JVal *m = sch->get("maximum")) if (m->kind == JVal::Num && v.num > m->num) errors.push_back(at + ": is greater than the maximum");
        }
        if (v.kind == JVal::Obj) {
            if (const JVal *req = sch->get("required"))
                if (req->kind == JVal::Arr)
                    for (auto &k : req->arr)
                        if (k.kind == JVal::Str && !v.get(k.str)) errors.push_back(at + ": \"" + k.str + "\" is a required property");
            if (const JVal *props = sch->get("properties"))
                if (props->kind == JVal::Obj)
                    for (auto &kv : props->obj)
                        if (const JVal *child = v.get(kv.first)) check(kv.second, *child, at + "." + kv.first);
        }
        if (v.kind == JVal::Arr)
            if (const JVal *items = sch->get("items"))
                if (items->kind == JVal::Obj)
                    for (size_t i = 0; i < v.arr.size(); i++) check(*items, v.arr[i], at + "[" + std::to_string(i) + "]");
    }
};

}  // namespace

extern "C" int dvt_stdin_from_json(const char *type, const char *json, size_t json_len, int auth_commitment, uint8_t **out, size_t *out_len,
                                   char **err_text) {
    if (err_text) *err_text = nullptr;
    auto bad = [&](const std::string &m) { if (err_text) *err_text = strdup(m.c_str()); return DVT_ERR_INPUT; };
    if (!type || !json || !out || !out_len) return bad("null argument");
    const Schema *s = schema_for(type);
    if (!s) return bad(std::string("unknown --type `") + type + "` (bad-share | finalization | bad-partial-key | bad-encrypted-share)");
    JParser jp{json, json + json_len, ""};
    JVal root;
    if (!jp.parse(&root)) return bad("JSON: " + jp.err);
    jp.ws();
    if (jp.p != jp.end) return bad("JSON: trailing characters");
    Cbor c;
    std::string err;
    if (!encode(*s, root, auth_commitment != 0, &c, &err, "$")) return bad(err);
    // SP1Stdin::write(&Vec<u8>) = bincode: u64 little-endian length, then the bytes
    const uint64_t n = c.out.size();
    uint8_t *buf = (uint8_t *)malloc(8 + n + 1);
    if (!buf) return bad("out of memory");
    for (int i = 0; i < 8; i++) buf[i] = (uint8_t)(n >> (8 * i));
    memcpy(buf + 8, c.out.data(), n);
    *out = buf;
    *out_len = 8 + n;
    return DVT_OK;
}

extern "C" int dvt_json_schema_validate(const char *schema, size_t schema_len, const char *json, size_t json_len, char **err_text) {
    if (err_text) *err_text = nullptr;
    auto bad = [&](const std::string &m) { if (err_text) *err_text = strdup(m.c_str()); return DVT_ERR_INPUT; };
    if (!schema || !json) return bad("null argument");
    JVal sch, doc;
    {
        JParser jp{schema, schema + schema_len, ""};
        if (!jp.parse(&sch)) return bad("Invalid JSON schema: " + jp.err);
    }
    {
        JParser jp{json, json + json_len, ""};
        if (!jp.parse(&doc)) return bad("Invalid JSON data: " + jp.err);
    }
    SchemaCheck c{sch, {}};
    c.check(sch, doc, "$");
    if (c.errors.empty()) return DVT_OK;
    std::string all;
    for (auto &e : c.errors) all += (all.empty() ? "" : "\n") + e;
    return bad(all);
}
