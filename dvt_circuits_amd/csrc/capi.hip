// C-ABI layer (include/dvt_prover.h) over the gfx950 kernels.  There is no CPU
// fallback anywhere in this file: without a HIP device every entry point that
// computes returns DVT_ERR_DEVICE.
#include "../../include/dvt_prover.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "kernels.h"

using namespace dvt;

struct dvt_prover {
    int device = 0;
    hipStream_t stream = nullptr;
    NttTables tabs;
    // ring of device/pinned-host words for small per-launch tables (column pointer lists)
    uint64_t *d_ring = nullptr, *h_ring = nullptr;
    size_t ring_words = 0, ring_pos = 0;
    std::string err;
    std::mutex mu;
};

static thread_local std::string g_create_err;

static int fail(dvt_prover *p, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (p) p->err = buf; else g_create_err = buf;
    return code;
}
#define HIP_TRY(p, expr)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(p, DVT_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static int cfg_int(const char *json, const char *key, int dflt) {
    if (!json) return dflt;
    std::string pat = std::string("\"") + key + "\"";
    const char *s = strstr(json, pat.c_str());
    if (!s) return dflt;
    s = strchr(s + pat.size(), ':');
    if (!s) return dflt;
    return atoi(s + 1);
}

extern "C" {

uint32_t dvt_abi_version(void) { return 1; }

int dvt_prover_create(const char *cfg_json, dvt_prover **out) {
    if (!out) return fail(nullptr, DVT_ERR_INPUT, "out == NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(nullptr, DVT_ERR_DEVICE, "no HIP device (%s); this library has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    int dev = cfg_int(cfg_json, "device", 0);
    if (dev < 0 || dev >= ndev) return fail(nullptr, DVT_ERR_INPUT, "device %d out of range (%d present)", dev, ndev);
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipSetDevice(dev));
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, DVT_ERR_DEVICE, "device %d is %s; this library is built for gfx950 only", dev, prop.gcnArchName);
    dvt_prover *p = new dvt_prover();
    p->device = dev;
    e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = ntt_tables_create(&p->tabs);
    p->ring_words = 1 << 17;
    if (e == hipSuccess) e = hipMalloc(&p->d_ring, p->ring_words * 8);
    if (e == hipSuccess) e = hipHostMalloc(&p->h_ring, p->ring_words * 8);
    if (e != hipSuccess) {
        fail(nullptr, DVT_ERR_DEVICE, "handle setup: %s", hipGetErrorString(e));
        dvt_prover_destroy(p);
        return DVT_ERR_DEVICE;
    }
    *out = p;
    return DVT_OK;
}

void dvt_prover_destroy(dvt_prover *p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    ntt_tables_destroy(&p->tabs);
    if (p->d_ring) (void)hipFree(p->d_ring);
    if (p->h_ring) (void)hipHostFree(p->h_ring);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

const char *dvt_last_error(const dvt_prover *p) { return p ? p->err.c_str() : g_create_err.c_str(); }

void dvt_free(void *ptr) { free(ptr); }

static hipStream_t pick(dvt_prover *p, void *stream) { return stream ? (hipStream_t)stream : p->stream; }

int dvt_sync(dvt_prover *p, void *stream) {
    if (!p) return DVT_ERR_INPUT;
    HIP_TRY(p, hipSetDevice(p->device));
    HIP_TRY(p, hipStreamSynchronize(pick(p, stream)));
    return DVT_OK;
}

int dvt_dev_to_internal(dvt_prover *p, void *stream, uint32_t *d, size_t n) {
    if (!p || (!d && n)) return fail(p, DVT_ERR_INPUT, "null argument");
    HIP_TRY(p, hipSetDevice(p->device));
    HIP_TRY(p, launch_to_internal(pick(p, stream), d, n));
    return DVT_OK;
}
int dvt_dev_from_internal(dvt_prover *p, void *stream, uint32_t *d, size_t n) {
    if (!p || (!d && n)) return fail(p, DVT_ERR_INPUT, "null argument");
    HIP_TRY(p, hipSetDevice(p->device));
    HIP_TRY(p, launch_from_internal(pick(p, stream), d, n));
    return DVT_OK;
}

int dvt_stage_coset_lde(dvt_prover *p, void *stream, uint32_t *d_in, uint32_t *d_out, uint32_t width, uint32_t log_n,
                        uint32_t shift_mode) {
    if (!p) return DVT_ERR_INPUT;
    if (width && (!d_in || !d_out)) return fail(p, DVT_ERR_INPUT, "null matrix");
    if (log_n > 22) return fail(p, DVT_ERR_INPUT, "log_n %u > 22", log_n);
    if (shift_mode > 2) return fail(p, DVT_ERR_INPUT, "shift_mode %u", shift_mode);
    HIP_TRY(p, hipSetDevice(p->device));
    HIP_TRY(p, launch_coset_lde(pick(p, stream), p->tabs, d_in, d_out, width, log_n, shift_mode));
    return DVT_OK;
}

size_t dvt_merkle_digest_words(const dvt_dev_matrix *mats, size_t n) {
    uint32_t mx = 0;
    for (size_t i = 0; i < n; i++) mx = std::max(mx, mats[i].log_height);
    return (((size_t)2 << mx) - 1) * 8;
}

// reserve `n` pointer slots in the ring, fill them on the host, queue the upload
static int ring_upload(dvt_prover *p, hipStream_t st, const std::vector<uint64_t> &ptrs, const uint32_t *const **d_out) {
    size_t n = ptrs.size();
    if (n > p->ring_words) return fail(p, DVT_ERR_INPUT, "too many columns (%zu)", n);
    if (p->ring_pos + n > p->ring_words) {
        HIP_TRY(p, hipStreamSynchronize(st));
        p->ring_pos = 0;
    }
    memcpy(p->h_ring + p->ring_pos, ptrs.data(), n * 8);
    HIP_TRY(p, hipMemcpyAsync(p->d_ring + p->ring_pos, p->h_ring + p->ring_pos, n * 8, hipMemcpyHostToDevice, st));
    *d_out = reinterpret_cast<const uint32_t *const *>(p->d_ring + p->ring_pos);
    p->ring_pos += n;
    return DVT_OK;
}

int dvt_stage_merkle_commit(dvt_prover *p, void *stream, const dvt_dev_matrix *mats, size_t n, uint32_t *d_digests) {
    if (!p) return DVT_ERR_INPUT;
    if (!mats || !n || !d_digests) return fail(p, DVT_ERR_INPUT, "null argument");
    std::lock_guard<std::mutex> lk(p->mu);
    HIP_TRY(p, hipSetDevice(p->device));
    hipStream_t st = pick(p, stream);
    uint32_t mx = 0;
    for (size_t i = 0; i < n; i++) {
        if (mats[i].log_height > 30) return fail(p, DVT_ERR_INPUT, "log_height too large");
        if (mats[i].width && !mats[i].d_data) return fail(p, DVT_ERR_INPUT, "null matrix data");
        mx = std::max(mx, mats[i].log_height);
    }
    uint32_t *prev = nullptr;
    for (uint32_t lh = mx + 1; lh-- > 0;) {
        std::vector<uint64_t> ptrs;
        for (size_t i = 0; i < n; i++)
            if (mats[i].log_height == lh)
                for (uint32_t c = 0; c < mats[i].width; c++)
                    ptrs.push_back((uint64_t)(uintptr_t)(mats[i].d_data + ((size_t)c << lh)));
        const uint32_t *const *d_cols = nullptr;
        if (!ptrs.empty()) {
            int rc = ring_upload(p, st, ptrs, &d_cols);
            if (rc) return rc;
        }
        if (lh == mx) {
            HIP_TRY(p, launch_merkle_leaves(st, d_cols, (uint32_t)ptrs.size(), lh, d_digests));
            prev = d_digests;
        } else {
            uint32_t *cur = prev + ((size_t)16 << lh);
            HIP_TRY(p, launch_merkle_level(st, prev, d_cols, (uint32_t)ptrs.size(), lh, cur));
            prev = cur;
        }
    }
    return DVT_OK;
}

int dvt_stage_poseidon2_permute(dvt_prover *p, void *stream, uint32_t *d_states, size_t n) {
    if (!p || (!d_states && n)) return fail(p, DVT_ERR_INPUT, "null argument");
    HIP_TRY(p, hipSetDevice(p->device));
    HIP_TRY(p, launch_poseidon2_permute(pick(p, stream), d_states, n));
    return DVT_OK;
}

}  // extern "C"
