// mdx_comm.hip — RCCL (xGMI) collectives of libmdx.so: one process per GPU, one
// small all-reduce of the accumulators at the end of an analysis.
//
// The reference gathers per-frame result arrays from worker processes and sums
// them on the parent (src/mdhelper/analysis/base.py:491-501,
// analysis/structure.py:841-844).  Here each rank accumulates on its own GPU and
// the accumulators meet in one ncclAllReduce; integer sums are order-independent,
// so the RDF counts stay bit-exact for any rank count.
#include "mdx_common.hpp"
#include "mdx_internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <rocfft/rocfft.h>

using namespace mdx;

#ifndef MDX_ROCM_ROOT
#define MDX_ROCM_ROOT "/opt/rocm"
#endif

struct mdx_comm {
    int dev = 0;
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    DeviceBuffer scratch;
};

#define MDX_NCCL(expr)                                                                    \
    do {                                                                                  \
        ncclResult_t _r = (expr);                                                         \
        if (_r != ncclSuccess)                                                            \
            return fail(MDX_ERR_RCCL, "%s failed: %s", #expr, ncclGetErrorString(_r));   \
    } while (0)


// the file a symbol this library calls is bound to (its own GOT entry, i.e. what the dynamic linker chose)
static const char *bound_file(const void *fn)
{
    Dl_info info;
    if (fn && dladdr(fn, &info) && info.dli_fname)
        return info.dli_fname;
    return "?";
}

extern "C" {

int mdx_runtime_info(char *buf, size_t bytes)
{
    MDX_REQUIRE(buf && bytes > 0, "NULL buffer");
    int hip_rt = 0, hip_drv = 0, rccl = 0;
    if (hipRuntimeGetVersion(&hip_rt) != hipSuccess)
        (void)hipGetLastError();
    if (hipDriverGetVersion(&hip_drv) != hipSuccess)
        (void)hipGetLastError();
    (void)ncclGetVersion(&rccl);
    char fftv[64] = "?";
    (void)rocfft_get_version_string(fftv, sizeof fftv);
    const int n = snprintf(buf, bytes,
                           "rocm_root=%s\nlibamdhip64=%s\nlibrocfft=%s\nlibrccl=%s\nlibmdx=%s\n"
                           "hip_runtime_version=%d\nhip_driver_version=%d\nrccl_version=%d\nrocfft_version=%s\n",
                           MDX_ROCM_ROOT, bound_file((const void *)&hipGetDeviceCount),
                           bound_file((const void *)&rocfft_setup), bound_file((const void *)&ncclGetVersion),
                           bound_file((const void *)&mdx_runtime_info), hip_rt, hip_drv, rccl, fftv);
    MDX_REQUIRE(n > 0 && size_t(n) < bytes, "buffer too small");
    return MDX_OK;
}

int mdx_comm_unique_id(unsigned char id[MDX_COMM_ID_BYTES])
{
    MDX_REQUIRE(id, "id is NULL");
    static_assert(sizeof(ncclUniqueId) == MDX_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId uid;
    MDX_NCCL(ncclGetUniqueId(&uid));
    memcpy(id, &uid, MDX_COMM_ID_BYTES);
    return MDX_OK;
}

int mdx_comm_init_rank(mdx_comm_t *out, int dev, const unsigned char id[MDX_COMM_ID_BYTES],
                       int rank, int world_size)
{
    MDX_REQUIRE(out && id, "NULL argument");
    MDX_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "bad rank/world_size");
    MDX_TRY(set_device(dev));
    mdx_comm *c = new mdx_comm();
    c->dev = dev;
    c->rank = rank;
    c->world = world_size;
    ncclUniqueId uid;
    memcpy(&uid, id, MDX_COMM_ID_BYTES);
    ncclResult_t r = ncclCommInitRank(&c->comm, world_size, uid, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(MDX_ERR_RCCL, "ncclCommInitRank failed: %s", ncclGetErrorString(r));
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        ncclCommDestroy(c->comm);
        delete c;
        return fail(MDX_ERR_HIP, "hipStreamCreate failed");
    }
    *out = c;
    return MDX_OK;
}

int mdx_comm_destroy(mdx_comm_t c)
{
    if (!c)
        return MDX_OK;
    (void)hipSetDevice(c->dev);
    if (c->stream)
        (void)hipStreamSynchronize(c->stream);
    if (c->comm)
        ncclCommDestroy(c->comm);
    c->scratch.release();
    if (c->stream)
        (void)hipStreamDestroy(c->stream);
    delete c;
    return MDX_OK;
}

static int allreduce_host(mdx_comm *c, void *host, int64_t n, ncclDataType_t dt, ncclRedOp_t op)
{
    MDX_TRY(set_device(c->dev));
    MDX_TRY(c->scratch.ensure(size_t(8) * n));
    MDX_HIP(hipMemcpyAsync(c->scratch.ptr, host, size_t(8) * n, hipMemcpyHostToDevice, c->stream));
    MDX_NCCL(ncclAllReduce(c->scratch.ptr, c->scratch.ptr, (size_t)n, dt, op, c->comm, c->stream));
    MDX_HIP(hipMemcpyAsync(host, c->scratch.ptr, size_t(8) * n, hipMemcpyDeviceToHost, c->stream));
    MDX_HIP(hipStreamSynchronize(c->stream));
    return MDX_OK;
}

int mdx_comm_count(mdx_comm_t c, int *count, int *rank, int *device)
{
    MDX_REQUIRE(c, "NULL communicator");
    int n = 0, r = 0, d = 0;
    MDX_NCCL(ncclCommCount(c->comm, &n));
    MDX_NCCL(ncclCommUserRank(c->comm, &r));
    MDX_NCCL(ncclCommCuDevice(c->comm, &d));
    if (count)
        *count = n;
    if (rank)
        *rank = r;
    if (device)
        *device = d;
    return MDX_OK;
}

int mdx_comm_barrier(mdx_comm_t c)
{
    MDX_REQUIRE(c, "NULL communicator");
    int64_t one = 1;
    return allreduce_host(c, &one, 1, ncclInt64, ncclSum);
}

int mdx_comm_allreduce_f64(mdx_comm_t c, double *host_inout, int64_t n, int op_max)
{
    MDX_REQUIRE(c && host_inout && n > 0, "bad argument");
    return allreduce_host(c, host_inout, n, ncclDouble, op_max ? ncclMax : ncclSum);
}

int mdx_comm_allreduce_i64(mdx_comm_t c, int64_t *host_inout, int64_t n)
{
    MDX_REQUIRE(c && host_inout && n > 0, "bad argument");
    return allreduce_host(c, host_inout, n, ncclInt64, ncclSum);
}

int mdx_rdf_allreduce(mdx_rdf_t h, mdx_comm_t c)
{
    MDX_REQUIRE(h && c, "NULL argument");
    unsigned long long *d_total = nullptr;
    hipStream_t s = nullptr;
    MDX_TRY(mdx_rdf_internal_total(h, &d_total, &s));
    const int n_bins = mdx_rdf_internal_nbins(h);
    MDX_NCCL(ncclAllReduce(d_total, d_total, (size_t)n_bins, ncclUint64, ncclSum, c->comm, s));
    MDX_TRY(mdx_rdf_internal_adopt_total(h));
    MDX_HIP(hipStreamSynchronize(s));
    return MDX_OK;
}

int mdx_sq_allreduce(mdx_sq_t h, mdx_comm_t c)
{
    MDX_REQUIRE(h && c, "NULL argument");
    double *d = nullptr;
    int64_t n = 0;
    hipStream_t s = nullptr;
    MDX_TRY(mdx_sq_internal_buffer(h, &d, &n, &s));
    MDX_NCCL(ncclAllReduce(d, d, (size_t)n, ncclDouble, ncclSum, c->comm, s));
    MDX_HIP(hipStreamSynchronize(s));
    return MDX_OK;
}

int mdx_msd_allreduce(mdx_msd_t h, mdx_comm_t c)
{
    MDX_REQUIRE(h && c, "NULL argument");
    double *da = nullptr, *db = nullptr;
    int64_t na = 0, nb = 0;
    hipStream_t s = nullptr;
    MDX_TRY(mdx_msd_internal_buffers(h, &da, &na, &db, &nb, &s));
    MDX_NCCL(ncclGroupStart());
    MDX_NCCL(ncclAllReduce(da, da, (size_t)na, ncclDouble, ncclSum, c->comm, s));
    MDX_NCCL(ncclAllReduce(db, db, (size_t)nb, ncclDouble, ncclSum, c->comm, s));
    MDX_NCCL(ncclGroupEnd());
    MDX_HIP(hipStreamSynchronize(s));
    return MDX_OK;
}

}  // extern "C"
