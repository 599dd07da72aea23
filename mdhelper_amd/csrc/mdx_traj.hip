// mdx_traj.hip — native trajectory ingest: AMBER NetCDF (CDF-1/CDF-2) and DCD readers that
// feed HBM directly (see mdx_traj.hpp).  SURVEY.md §8f row 4; the NetCDF variable names and
// container format follow /root/reference/src/mdhelper/openmm/file.py:49-52, 160-188.
#include "mdx_traj.hpp"
#include "mdx_internal.hpp"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <string>
#include <thread>

namespace mdx {

namespace {

inline uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }
inline uint64_t bswap64(uint64_t v) { return __builtin_bswap64(v); }

// ----------------------------------------------------------------- NetCDF classic header
//
// header   = magic numrecs dim_list gatt_list var_list            (all integers big-endian)
// magic    = 'C' 'D' 'F' version      version 1: 32-bit offsets, 2: 64-bit offsets
// list     = ABSENT (two zero words) | tag nelems items          tags: 10 dim, 11 var, 12 att
// dim      = name length                                           length 0: the record dimension
// att      = name nc_type nelems values(padded to 4)
// var      = name ndims dimid* att_list nc_type vsize begin        begin 4 or 8 bytes
// Record variables (first dimension = record dimension) are interleaved record by record in
// definition order; each slice is padded to 4 bytes unless it is the only record variable.
struct Cursor {
    const uint8_t *p;
    size_t n, at = 0;
    bool short_read = false;
    bool need(size_t k)
    {
        if (at + k > n) {
            short_read = true;
            return false;
        }
        return true;
    }
    uint32_t u32()
    {
        if (!need(4))
            return 0;
        uint32_t v;
        memcpy(&v, p + at, 4);
        at += 4;
        return bswap32(v);
    }
    uint64_t u64()
    {
        if (!need(8))
            return 0;
        uint64_t v;
        memcpy(&v, p + at, 8);
        at += 8;
        return bswap64(v);
    }
    std::string name()
    {
        uint32_t len = u32();
        size_t padded = (size_t(len) + 3) & ~size_t(3);
        if (!need(padded))
            return std::string();
        std::string s(reinterpret_cast<const char *>(p + at), len);
        at += padded;
        return s;
    }
    void skip(size_t k)
    {
        if (need(k))
            at += k;
    }
};

int nc_type_size(uint32_t t)
{
    switch (t) {
    case 1: case 2: return 1;   // byte, char
    case 3: return 2;           // short
    case 4: case 5: return 4;   // int, float
    case 6: return 8;           // double
    default: return 0;
    }
}

struct NcAttr {
    std::string name;
    uint32_t type = 0, nelems = 0;
    size_t value_at = 0;   // offset of the values in the header buffer
};

struct NcVar {
    std::string name;
    std::vector<uint32_t> dims;
    std::vector<NcAttr> atts;
    uint32_t type = 0;
    uint64_t begin = 0;
    bool record = false;
    int64_t slice_bytes = 0;   // one record's worth (record var) or the whole variable
};

bool parse_atts(Cursor &c, std::vector<NcAttr> &out)
{
    uint32_t tag = c.u32(), n = c.u32();
    if (c.short_read)
        return false;
    if (tag == 0 && n == 0)
        return true;
    if (tag != 12)
        return false;
    for (uint32_t i = 0; i < n && !c.short_read; ++i) {
        NcAttr a;
        a.name = c.name();
        a.type = c.u32();
        a.nelems = c.u32();
        a.value_at = c.at;
        int ts = nc_type_size(a.type);
        if (!c.short_read && ts == 0)
            return false;
        c.skip((size_t(a.nelems) * ts + 3) & ~size_t(3));
        out.push_back(a);
    }
    return !c.short_read;
}

double attr_as_double(const std::vector<uint8_t> &head, const NcAttr &a)
{
    if (a.nelems < 1)
        return 1.0;
    if (a.type == 6) {
        uint64_t v;
        memcpy(&v, head.data() + a.value_at, 8);
        v = bswap64(v);
        double d;
        memcpy(&d, &v, 8);
        return d;
    }
    if (a.type == 5) {
        uint32_t v;
        memcpy(&v, head.data() + a.value_at, 4);
        v = bswap32(v);
        float f;
        memcpy(&f, &v, 4);
        return f;
    }
    if (a.type == 4) {
        uint32_t v;
        memcpy(&v, head.data() + a.value_at, 4);
        return (double)(int32_t)bswap32(v);
    }
    return 1.0;
}

double load_scalar(const uint8_t *p, int nc_type, bool swap)
{
    if (nc_type == 6) {
        uint64_t v;
        memcpy(&v, p, 8);
        if (swap)
            v = bswap64(v);
        double d;
        memcpy(&d, &v, 8);
        return d;
    }
    uint32_t v;
    memcpy(&v, p, 4);
    if (swap)
        v = bswap32(v);
    float f;
    memcpy(&f, &v, 4);
    return f;
}

}  // namespace

int Trajectory::read_at(int64_t offset, void *dst, size_t bytes) const
{
    uint8_t *out = static_cast<uint8_t *>(dst);
    while (bytes) {
        ssize_t got = pread(fd, out, bytes, offset);
        if (got < 0) {
            if (errno == EINTR)
                continue;
            return fail(MDX_ERR_IO, "trajectory read failed at byte %lld: %s", (long long)offset,
                        strerror(errno));
        }
        if (got == 0)
            return fail(MDX_ERR_IO, "trajectory file is truncated (wanted byte %lld of %lld)",
                        (long long)offset, (long long)file_bytes);
        out += got;
        offset += got;
        bytes -= size_t(got);
    }
    return MDX_OK;
}

int Trajectory::parse_netcdf(const std::vector<uint8_t> &head, bool &need_more)
{
    Cursor c{head.data(), head.size()};
    c.skip(3);
    if (!c.need(1))
        return need_more = true, MDX_OK;
    const int version = head[3];
    c.at = 4;
    if (version == 5)
        return fail(MDX_ERR_UNSUPPORTED, "NetCDF CDF-5 (64-bit data) containers are not supported; "
                    "AMBER trajectories use CDF-1 or CDF-2 (NETCDF3_64BIT_OFFSET)");
    if (version != 1 && version != 2)
        return fail(MDX_ERR_INVALID_VALUE, "not a NetCDF classic file (version byte %d)", version);
    const uint32_t numrecs = c.u32();

    // dimensions
    std::vector<std::string> dim_names;
    std::vector<uint32_t> dim_len;
    int rec_dim = -1;
    {
        uint32_t tag = c.u32(), n = c.u32();
        if (!c.short_read && !(tag == 0 && n == 0)) {
            if (tag != 10)
                return fail(MDX_ERR_INVALID_VALUE, "NetCDF header: dimension list expected");
            for (uint32_t i = 0; i < n && !c.short_read; ++i) {
                dim_names.push_back(c.name());
                dim_len.push_back(c.u32());
                if (!c.short_read && dim_len.back() == 0)
                    rec_dim = (int)i;
            }
        }
    }
    std::vector<NcAttr> gatts;
    if (!c.short_read && !parse_atts(c, gatts) && !c.short_read)
        return fail(MDX_ERR_INVALID_VALUE, "NetCDF header: malformed global attribute list");

    std::vector<NcVar> vars;
    if (!c.short_read) {
        uint32_t tag = c.u32(), n = c.u32();
        if (!c.short_read && !(tag == 0 && n == 0)) {
            if (tag != 11)
                return fail(MDX_ERR_INVALID_VALUE, "NetCDF header: variable list expected");
            for (uint32_t i = 0; i < n && !c.short_read; ++i) {
                NcVar v;
                v.name = c.name();
                uint32_t nd = c.u32();
                for (uint32_t k = 0; k < nd && !c.short_read; ++k)
                    v.dims.push_back(c.u32());
                if (!parse_atts(c, v.atts) && !c.short_read)
                    return fail(MDX_ERR_INVALID_VALUE, "NetCDF header: malformed attributes of '%s'",
                                v.name.c_str());
                v.type = c.u32();
                c.u32();   // vsize: recomputed below (it saturates for large variables)
                v.begin = (version == 2) ? c.u64() : c.u32();
                if (c.short_read)
                    break;
                int64_t bytes = nc_type_size(v.type);
                if (bytes == 0)
                    return fail(MDX_ERR_INVALID_VALUE, "NetCDF header: variable '%s' has unknown type",
                                v.name.c_str());
                for (size_t k = 0; k < v.dims.size(); ++k) {
                    if (v.dims[k] >= dim_len.size())
                        return fail(MDX_ERR_INVALID_VALUE, "NetCDF header: bad dimension id");
                    if (k == 0 && (int)v.dims[k] == rec_dim)
                        v.record = true;
                    else
                        bytes *= dim_len[v.dims[k]];
                }
                v.slice_bytes = bytes;
                vars.push_back(v);
            }
        }
    }
    if (c.short_read) {
        need_more = true;
        return MDX_OK;
    }

    // record size
    int n_rec_vars = 0;
    for (const NcVar &v : vars)
        n_rec_vars += v.record;
    int64_t rec_size = 0;
    for (const NcVar &v : vars)
        if (v.record)
            rec_size += (n_rec_vars == 1) ? v.slice_bytes : ((v.slice_bytes + 3) & ~int64_t(3));

    auto find = [&](const char *nm) -> const NcVar * {
        for (const NcVar &v : vars)
            if (v.name == nm)
                return &v;
        return nullptr;
    };
    for (const NcAttr &a : gatts)
        if (a.name == "Conventions") {
            std::string val(reinterpret_cast<const char *>(head.data() + a.value_at), a.nelems);
            if (val.find("AMBERRESTART") != std::string::npos)
                return fail(MDX_ERR_UNSUPPORTED, "AMBER NetCDF restart files hold a single frame "
                            "without a frame dimension; pass a trajectory (Conventions = AMBER)");
        }
    const NcVar *xyz = find("coordinates");
    if (!xyz)
        return fail(MDX_ERR_INVALID_VALUE, "NetCDF file has no 'coordinates' variable");
    if (!xyz->record || xyz->dims.size() != 3)
        return fail(MDX_ERR_INVALID_VALUE, "'coordinates' must be (frame, atom, spatial)");
    if (xyz->type != 5)
        return fail(MDX_ERR_UNSUPPORTED, "'coordinates' must be float32 (AMBER convention)");
    if (dim_len[xyz->dims[2]] != 3)
        return fail(MDX_ERR_INVALID_VALUE, "'coordinates' must have 3 spatial components");
    for (const NcAttr &a : xyz->atts)
        if (a.name == "scale_factor")
            coord_scale = attr_as_double(head, a);
    if (coord_scale != 1.0)
        return fail(MDX_ERR_UNSUPPORTED, "'coordinates' scale_factor %g is not supported",
                    coord_scale);

    format = TRAJ_NETCDF;
    const uint16_t probe = 1;
    swap = *reinterpret_cast<const uint8_t *>(&probe) == 1;   // file is big-endian
    n_atoms = dim_len[xyz->dims[1]];
    frame_stride = rec_size;
    coord_first = (int64_t)xyz->begin;
    plane_stride = 0;
    int64_t rec_begin = coord_first;
    for (const NcVar &v : vars)
        if (v.record)
            rec_begin = std::min<int64_t>(rec_begin, (int64_t)v.begin);
    if (numrecs == 0xFFFFFFFFu)   // "streaming": derive from the file size
        n_frames = rec_size > 0 ? (file_bytes - rec_begin) / rec_size : 0;
    else
        n_frames = numrecs;
    // every declared record must be present (the last slice may lack its padding)
    if (rec_begin + n_frames * rec_size - 3 > file_bytes)
        return fail(MDX_ERR_IO, "NetCDF file is truncated: %lld frames declared, %lld bytes "
                    "needed, %lld present", (long long)n_frames,
                    (long long)(rec_begin + n_frames * rec_size), (long long)file_bytes);
    const NcVar *len = find("cell_lengths"), *ang = find("cell_angles"), *tim = find("time");
    if (len && len->record && len->slice_bytes == 3 * nc_type_size(len->type)
        && (len->type == 5 || len->type == 6)) {
        has_box = true;
        cell_first = (int64_t)len->begin;
        cell_type = (int)len->type;
        if (ang && ang->record && ang->slice_bytes == 3 * nc_type_size(ang->type)
            && (ang->type == 5 || ang->type == 6)) {
            angle_first = (int64_t)ang->begin;
            angle_type = (int)ang->type;
        }
    }
    if (tim && tim->record && (tim->type == 5 || tim->type == 6)
        && tim->slice_bytes == nc_type_size(tim->type)) {
        has_time = true;
        time_first = (int64_t)tim->begin;
        time_type = (int)tim->type;
    }
    return MDX_OK;
}

// ----------------------------------------------------------------------------- DCD header
//
// Fortran unformatted records (int32 length, payload, int32 length):
//   [84]  'CORD' icntrl[20]   icntrl: 0 NSET, 1 ISTART, 2 NSAVC, 8 NFIXED, 9 DELTA (float32),
//                                     10 unit-cell flag, 11 fourth-dimension flag, 19 version
//   [4 + 80 k]  NTITLE, titles
//   [4]   NATOM
//   per frame: [48] unit cell (A, gamma, B, beta, alpha, C as doubles) when flagged,
//              [4 N] x, [4 N] y, [4 N] z, ([4 N] w when the fourth dimension is flagged)
int Trajectory::parse_dcd(const std::vector<uint8_t> &head, bool &need_more)
{
    if (head.size() < 92) {
        need_more = true;
        return MDX_OK;
    }
    auto rd32 = [&](size_t at) {
        uint32_t v;
        memcpy(&v, head.data() + at, 4);
        return swap ? bswap32(v) : v;
    };
    uint32_t first;
    memcpy(&first, head.data(), 4);
    if (first == 84u)
        swap = false;
    else if (bswap32(first) == 84u)
        swap = true;
    else
        return fail(MDX_ERR_INVALID_VALUE, "not a DCD file (first record length is not 84)");
    if (memcmp(head.data() + 4, "CORD", 4) != 0)
        return fail(MDX_ERR_INVALID_VALUE, "DCD header: 'CORD' tag missing (velocity files are "
                    "not trajectories)");
    int32_t icntrl[20];
    for (int i = 0; i < 20; ++i)
        icntrl[i] = (int32_t)rd32(8 + 4 * i);
    if (rd32(88) != 84u)
        return fail(MDX_ERR_INVALID_VALUE, "DCD header: first record is not closed");
    if (icntrl[8] != 0)
        return fail(MDX_ERR_UNSUPPORTED, "DCD files with fixed atoms (NFIXED = %d) are not "
                    "supported", icntrl[8]);
    const bool charmm = icntrl[19] != 0;
    const bool cell = charmm && icntrl[10] != 0;
    const bool dim4 = charmm && icntrl[11] != 0;
    if (charmm) {
        uint32_t bits = (uint32_t)icntrl[9];
        float f;
        memcpy(&f, &bits, 4);
        dcd_delta = f;
    } else {   // X-PLOR: DELTA is a double over words 9 and 10
        uint64_t bits;
        memcpy(&bits, head.data() + 8 + 4 * 9, 8);
        if (swap)
            bits = bswap64(bits);
        memcpy(&dcd_delta, &bits, 8);
    }
    dcd_istart = icntrl[1];
    dcd_nsavc = icntrl[2] > 0 ? icntrl[2] : 1;
    // title record
    size_t at = 92;
    if (head.size() < at + 4) {
        need_more = true;
        return MDX_OK;
    }
    const uint32_t title_len = rd32(at);
    if (title_len < 4 || (title_len - 4) % 80 != 0)
        return fail(MDX_ERR_INVALID_VALUE, "DCD header: malformed title record (%u bytes)",
                    title_len);
    at += 4 + size_t(title_len) + 4;
    if (head.size() < at + 12) {
        need_more = true;
        return MDX_OK;
    }
    if (rd32(at) != 4u || rd32(at + 8) != 4u)
        return fail(MDX_ERR_INVALID_VALUE, "DCD header: atom-count record expected");
    n_atoms = (int32_t)rd32(at + 4);
    if (n_atoms <= 0)
        return fail(MDX_ERR_INVALID_VALUE, "DCD header: %lld atoms", (long long)n_atoms);
    at += 12;
    format = TRAJ_DCD;
    const int64_t plane = 4 * n_atoms + 8;
    frame_stride = (cell ? 56 : 0) + plane * (dim4 ? 4 : 3);
    const int64_t frame0 = (int64_t)at;
    has_box = cell;
    cell_first = cell ? frame0 + 4 : -1;
    coord_first = frame0 + (cell ? 56 : 0) + 4;
    plane_stride = plane;
    has_time = true;
    // NSET in the header is not always kept up to date by writers; trust the file size
    const int64_t by_size = (file_bytes - frame0) / frame_stride;
    n_frames = icntrl[0] > 0 ? std::min<int64_t>(icntrl[0], by_size) : by_size;
    return MDX_OK;
}

int Trajectory::open(const char *path)
{
    fd = ::open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0)
        return fail(MDX_ERR_IO, "cannot open '%s': %s", path, strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0)
        return fail(MDX_ERR_IO, "cannot stat '%s': %s", path, strerror(errno));
    file_bytes = st.st_size;
    if (file_bytes < 8)
        return fail(MDX_ERR_INVALID_VALUE, "'%s' is too short to be a trajectory", path);
    // the header is read in growing prefixes until the parser has seen all of it
    for (size_t want = 64 << 10;; want *= 4) {
        std::vector<uint8_t> head(std::min<size_t>(want, (size_t)file_bytes));
        MDX_TRY(read_at(0, head.data(), head.size()));
        bool need_more = false;
        int rc;
        if (memcmp(head.data(), "CDF", 3) == 0)
            rc = parse_netcdf(head, need_more);
        else if (memcmp(head.data(), "\x89HDF", 4) == 0)
            return fail(MDX_ERR_UNSUPPORTED, "'%s' is a NetCDF-4/HDF5 container; AMBER "
                        "trajectories are NetCDF-3 (ncdump -k says 'classic' or '64-bit offset')",
                        path);
        else
            rc = parse_dcd(head, need_more);
        MDX_TRY(rc);
        if (!need_more)
            break;
        if (head.size() >= (size_t)file_bytes)
            return fail(MDX_ERR_IO, "'%s': header runs past the end of the file", path);
    }
    return MDX_OK;
}

void Trajectory::close()
{
    if (dev >= 0) {
        (void)hipSetDevice(dev);
        {
            // what this trajectory queued on the shared ring's stream reads d_raw
            HostStager &ring = device_stager(dev);
            std::lock_guard<std::mutex> guard(ring.lock);
            if (ring.io)
                (void)hipStreamSynchronize(ring.io);
        }
        d_raw.release();
        dev = -1;
    }
    if (fd >= 0)
        ::close(fd);
    fd = -1;
}

// raw coordinate bytes of the listed frames, 12 N per frame, in file order and layout
int Trajectory::fill_raw(const int64_t *frames, int64_t n, uint8_t *dst, HostWorkers *workers) const
{
    const int64_t per_frame = 12 * n_atoms;
    // pieces of about 1 MiB (a frame, or a slice of a large frame's bytes / planes): enough of them
    // for every reader thread whatever the frame size
    const int64_t unit = format == TRAJ_NETCDF ? per_frame : 4 * n_atoms;     // one contiguous run on disk
    const int64_t units_per_frame = format == TRAJ_NETCDF ? 1 : 3;
    const int64_t split = std::max<int64_t>(1, std::min<int64_t>(64, unit >> 20));
    const int64_t n_tasks = n * units_per_frame * split;
    // Workers only record (status, message): mdx_last_error() is thread-local, so the failing
    // thread's own fail() would be invisible to the caller; the message is restated on the calling
    // thread after the join.
    std::mutex err_lock;
    int first_rc = MDX_OK;
    std::string first_msg;
    const std::function<void(int)> work = [&](int task) {
        const int64_t i = task / (units_per_frame * split);
        const int64_t r = task - i * units_per_frame * split;
        const int64_t k = r / split, part = r - k * split;
        const int64_t lo = unit * part / split, hi = unit * (part + 1) / split;
        const int64_t at = coord_first + frames[i] * frame_stride + k * plane_stride + lo;
        int rc = read_at(at, dst + i * per_frame + k * unit + lo, size_t(hi - lo));
        if (rc != MDX_OK) {
            std::lock_guard<std::mutex> lk(err_lock);
            if (first_rc == MDX_OK) {
                first_rc = rc;
                first_msg = mdx_last_error();   // this thread's message
            }
        }
    };
    if (n_tasks > (int64_t(1) << 30))
        return fail(MDX_ERR_INVALID_VALUE, "too many frames in one read");
    if (workers && n * per_frame >= (int64_t(4) << 20)) {
        workers->parallel_for(int(n_tasks), work);
    } else {
        for (int64_t t = 0; t < n_tasks && first_rc == MDX_OK; ++t)
            work(int(t));
    }
    if (first_rc != MDX_OK)
        return fail(first_rc, "%s", first_msg.empty() ? "trajectory read failed" : first_msg.c_str());
    return MDX_OK;
}

static int check_frames(const Trajectory &t, const int64_t *frames, int64_t n)
{
    MDX_REQUIRE(n >= 0, "negative frame count");
    MDX_REQUIRE(n == 0 || frames, "NULL frame list");
    for (int64_t i = 0; i < n; ++i)
        if (frames[i] < 0 || frames[i] >= t.n_frames)
            return fail(MDX_ERR_INVALID_VALUE, "frame %lld out of range [0, %lld)",
                        (long long)frames[i], (long long)t.n_frames);
    return MDX_OK;
}

int Trajectory::read_positions(const int64_t *frames, int64_t n, float *out) const
{
    MDX_TRY(check_frames(*this, frames, n));
    if (n == 0)
        return MDX_OK;
    MDX_REQUIRE(out, "NULL output");
    const int64_t N = n_atoms;
    if (format == TRAJ_NETCDF) {
        HostWorkers readers;
        if (n * N * 12 >= (int64_t(8) << 20))
            readers.start(4);
        MDX_TRY(fill_raw(frames, n, reinterpret_cast<uint8_t *>(out), &readers));
        if (swap) {
            uint32_t *w = reinterpret_cast<uint32_t *>(out);
            for (int64_t i = 0; i < n * N * 3; ++i)
                w[i] = bswap32(w[i]);
        }
        return MDX_OK;
    }
    std::vector<uint32_t> planes(size_t(3) * N);
    for (int64_t f = 0; f < n; ++f) {
        MDX_TRY(fill_raw(frames + f, 1, reinterpret_cast<uint8_t *>(planes.data())));
        uint32_t *o = reinterpret_cast<uint32_t *>(out + f * N * 3);
        for (int64_t a = 0; a < N; ++a)
            for (int k = 0; k < 3; ++k) {
                uint32_t v = planes[size_t(k) * N + a];
                o[a * 3 + k] = swap ? bswap32(v) : v;
            }
    }
    return MDX_OK;
}

int Trajectory::read_boxes(const int64_t *frames, int64_t n, float *boxes6) const
{
    MDX_TRY(check_frames(*this, frames, n));
    if (n == 0)
        return MDX_OK;
    MDX_REQUIRE(boxes6, "NULL output");
    if (!has_box)
        return fail(MDX_ERR_STATE, "the trajectory holds no unit-cell information");
    // one or two small reads per frame, a record apart on disk: shared out over a few threads for long
    // lists (12 000 reads for 6 000 frames were 5 % of an RDF over a file at the kernel's rate)
    std::mutex err_lock;
    int first_rc = MDX_OK;
    std::string first_msg;
    const std::function<void(int)> one = [&](int task) {
        const int64_t lo = int64_t(task) * 256, hi = std::min<int64_t>(n, lo + 256);
        for (int64_t i = lo; i < hi; ++i) {
            const int rc = read_box(frames[i], boxes6 + 6 * i);
            if (rc != MDX_OK) {
                std::lock_guard<std::mutex> lk(err_lock);
                if (first_rc == MDX_OK) {
                    first_rc = rc;
                    first_msg = mdx_last_error();
                }
                return;
            }
        }
    };
    const int tasks = (int)ceil_div(n, 256);
    if (n >= 2048) {
        HostWorkers readers;
        readers.start(8);
        readers.parallel_for(tasks, one);
    } else {
        for (int t = 0; t < tasks && first_rc == MDX_OK; ++t)
            one(t);
    }
    if (first_rc != MDX_OK)
        return fail(first_rc, "%s", first_msg.empty() ? "trajectory read failed" : first_msg.c_str());
    return MDX_OK;
}

int Trajectory::read_box(int64_t frame, float *b) const
{
    uint8_t raw[48];
    if (format == TRAJ_NETCDF) {
        const int ls = nc_type_size(cell_type);
        MDX_TRY(read_at(cell_first + frame * frame_stride, raw, size_t(3) * ls));
        for (int k = 0; k < 3; ++k)
            b[k] = (float)load_scalar(raw + k * ls, cell_type, swap);
        b[3] = b[4] = b[5] = 90.0f;
        if (angle_first >= 0) {
            const int as = nc_type_size(angle_type);
            MDX_TRY(read_at(angle_first + frame * frame_stride, raw, size_t(3) * as));
            for (int k = 0; k < 3; ++k)
                b[3 + k] = (float)load_scalar(raw + k * as, angle_type, swap);
        }
    } else {
        MDX_TRY(read_at(cell_first + frame * frame_stride, raw, 48));
        double u[6];
        for (int k = 0; k < 6; ++k)
            u[k] = load_scalar(raw + 8 * k, 6, swap);
        // stored as A, gamma, B, beta, alpha, C; CHARMM and NAMD store the angle cosines
        if (std::fabs(u[1]) <= 1.0 && std::fabs(u[3]) <= 1.0 && std::fabs(u[4]) <= 1.0) {
            for (int k : {1, 3, 4})
                u[k] = 90.0 - std::asin(u[k]) * 90.0 / M_PI_2;
        }
        b[0] = (float)u[0];
        b[1] = (float)u[2];
        b[2] = (float)u[5];
        b[3] = (float)u[4];
        b[4] = (float)u[3];
        b[5] = (float)u[1];
    }
    return MDX_OK;
}

int Trajectory::read_times(const int64_t *frames, int64_t n, double *times) const
{
    MDX_TRY(check_frames(*this, frames, n));
    if (n == 0)
        return MDX_OK;
    MDX_REQUIRE(times, "NULL output");
    if (!has_time)
        return fail(MDX_ERR_STATE, "the trajectory holds no time variable");
    for (int64_t i = 0; i < n; ++i) {
        if (format == TRAJ_NETCDF) {
            uint8_t raw[8];
            MDX_TRY(read_at(time_first + frames[i] * frame_stride, raw,
                            (size_t)nc_type_size(time_type)));
            times[i] = load_scalar(raw, time_type, swap);
        } else {
            // AKMA time unit -> ps
            times[i] = double(dcd_istart + frames[i] * dcd_nsavc) * dcd_delta * 4.888821e-2;
        }
    }
    return MDX_OK;
}

// ------------------------------------------------------------------------- device pipeline

// out[f][s][k] = fix(raw[f][src(s, k)])  with  src = 3 a + k (NetCDF) or k N + a (DCD planes),
// a = index ? index[s] : s.  One thread per output float; reads of the un-gathered NetCDF
// layout and all writes are contiguous.
__global__ __launch_bounds__(256) void traj_unpack_kernel(const uint32_t *__restrict__ raw,
                                                          const int *__restrict__ index,
                                                          uint32_t *__restrict__ out, int64_t n_atoms,
                                                          int64_t n_sel, int64_t n_frames, int planes,
                                                          int swap)
{
    const int64_t per_out = 3 * n_sel;
    const int64_t total = per_out * n_frames;
    for (int64_t e = blockIdx.x * int64_t(256) + threadIdx.x; e < total;
         e += int64_t(gridDim.x) * 256) {
        const int64_t f = e / per_out;
        const int64_t r = e - f * per_out;
        const int64_t s = r / 3;
        const int k = int(r - 3 * s);
        const int64_t a = index ? index[s] : s;
        const int64_t src = planes ? k * n_atoms + a : 3 * a + k;
        uint32_t v = raw[f * 3 * n_atoms + src];
        out[e] = swap ? __builtin_bswap32(v) : v;
    }
}

static int check_selections(const Trajectory &t, const TrajSelection *sel, int n_sel)
{
    MDX_REQUIRE(n_sel >= 1 && sel, "no selection given");
    for (int i = 0; i < n_sel; ++i)
        MDX_REQUIRE(sel[i].n_sel >= 0 && (sel[i].d_index || sel[i].n_sel <= t.n_atoms),
                    "selection %d is larger than the trajectory", i);
    return MDX_OK;
}

static void launch_unpack(const Trajectory &t, hipStream_t stream, const void *d_raw, int64_t f0, int64_t nf,
                          const TrajSelection *sel, int n_sel)
{
    for (int i = 0; i < n_sel; ++i) {
        if (sel[i].n_sel == 0)
            continue;
        const int64_t total = 3 * sel[i].n_sel * nf;
        const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(total, 256), 65536);
        hipLaunchKernelGGL(traj_unpack_kernel, dim3(grid), dim3(256), 0, stream,
                           static_cast<const uint32_t *>(d_raw), sel[i].d_index,
                           reinterpret_cast<uint32_t *>(sel[i].d_out + f0 * sel[i].n_sel * 3), t.n_atoms,
                           sel[i].n_sel, nf, t.format == TRAJ_DCD ? 1 : 0, t.swap ? 1 : 0);
    }
}

int Trajectory::stage_async(int device, hipStream_t consumer, const int64_t *frames, int64_t n,
                            const TrajSelection *sel, int n_sel)
{
    MDX_TRY(check_frames(*this, frames, n));
    MDX_TRY(check_selections(*this, sel, n_sel));
    if (n == 0)
        return MDX_OK;
    if (dev >= 0 && dev != device)
        return fail(MDX_ERR_STATE, "trajectory handle is bound to device %d", dev);
    MDX_TRY(set_device(device));
    dev = device;
    HostStager &ring = device_stager(device);
    std::lock_guard<std::mutex> guard(ring.lock);
    const int64_t per_frame = 12 * n_atoms;
    const int64_t chunk = std::max<int64_t>(1, (int64_t(16) << 20) / per_frame);
    MDX_TRY(ring.ensure(device, size_t(chunk * per_frame)));
    MDX_TRY(d_raw.ensure(size_t(chunk * per_frame)));
    // the outputs may still be read by kernels the consumer queued earlier
    MDX_TRY(ring.after(consumer));
    int rc = MDX_OK;
    for (int64_t f0 = 0; f0 < n && rc == MDX_OK; f0 += chunk) {
        const int64_t nf = std::min(chunk, n - f0);
        int b;
        void *host;
        if ((rc = ring.acquire(&b, &host)) != MDX_OK)
            break;
        if ((rc = fill_raw(frames + f0, nf, static_cast<uint8_t *>(host), &ring.workers)) != MDX_OK)
            break;
        // the ring's stream runs in order: this copy into d_raw follows the unpack kernels of the previous chunk
        if ((rc = ring.send(b, d_raw.ptr, size_t(nf * per_frame))) != MDX_OK)
            break;
        launch_unpack(*this, ring.io, d_raw.ptr, f0, nf, sel, n_sel);
        if (hipGetLastError() != hipSuccess)
            rc = fail(MDX_ERR_HIP, "trajectory unpack kernel launch failed");
    }
    // on every path — a failed read included — the consumer is ordered behind what did get queued,
    // so the outputs are never written behind its back; nothing is left recorded on its stream
    const int rc2 = ring.finish(consumer);
    return rc != MDX_OK ? rc : rc2;
}

int Trajectory::stage_raw_async(int device, hipStream_t consumer, const int64_t *frames, int64_t n,
                                void *d_raw_out)
{
    MDX_TRY(check_frames(*this, frames, n));
    if (n == 0)
        return MDX_OK;
    MDX_REQUIRE(d_raw_out, "NULL output");
    if (dev >= 0 && dev != device)
        return fail(MDX_ERR_STATE, "trajectory handle is bound to device %d", dev);
    MDX_TRY(set_device(device));
    dev = device;
    HostStager &ring = device_stager(device);
    std::lock_guard<std::mutex> guard(ring.lock);
    const int64_t per_frame = 12 * n_atoms;
    const int64_t chunk = std::max<int64_t>(1, (int64_t(16) << 20) / per_frame);
    MDX_TRY(ring.ensure(device, size_t(chunk * per_frame)));
    int rc = MDX_OK;
    for (int64_t f0 = 0; f0 < n && rc == MDX_OK; f0 += chunk) {
        const int64_t nf = std::min(chunk, n - f0);
        int b;
        void *host;
        if ((rc = ring.acquire(&b, &host)) != MDX_OK)
            break;
        if ((rc = fill_raw(frames + f0, nf, static_cast<uint8_t *>(host), &ring.workers)) != MDX_OK)
            break;
        rc = ring.send(b, static_cast<uint8_t *>(d_raw_out) + f0 * per_frame, size_t(nf * per_frame));
    }
    const int rc2 = ring.finish(consumer);
    return rc != MDX_OK ? rc : rc2;
}

int Trajectory::unpack_async(hipStream_t stream, const void *d_raw_in, int64_t n, const TrajSelection *sel,
                             int n_sel) const
{
    MDX_TRY(check_selections(*this, sel, n_sel));
    if (n == 0)
        return MDX_OK;
    launch_unpack(*this, stream, d_raw_in, 0, n, sel, n_sel);
    MDX_HIP(hipGetLastError());
    return MDX_OK;
}

}  // namespace mdx

// ---------------------------------------------------------------------------------- C-ABI

struct mdx_traj {
    mdx::Trajectory t;
    hipStream_t stream = nullptr;   // for mdx_traj_load_device
};

mdx::Trajectory *mdx_traj_internal(mdx_traj_t h) { return h ? &h->t : nullptr; }

extern "C" {

using namespace mdx;

int mdx_traj_open(mdx_traj_t *out, const char *path)
{
    MDX_REQUIRE(out && path, "NULL argument");
    *out = nullptr;
    mdx_traj *h = new mdx_traj();
    int rc = h->t.open(path);
    if (rc != MDX_OK) {
        h->t.close();
        delete h;
        return rc;
    }
    *out = h;
    return MDX_OK;
}

int mdx_traj_close(mdx_traj_t h)
{
    if (!h)
        return MDX_OK;
    if (h->stream) {
        (void)hipSetDevice(h->t.dev);
        (void)hipStreamSynchronize(h->stream);
    }
    h->t.close();
    if (h->stream)
        (void)hipStreamDestroy(h->stream);
    delete h;
    return MDX_OK;
}

int mdx_traj_info(mdx_traj_t h, int64_t *n_frames, int64_t *n_atoms, int *has_box, int *has_time,
                  int *format)
{
    MDX_REQUIRE(h, "NULL handle");
    if (n_frames) *n_frames = h->t.n_frames;
    if (n_atoms) *n_atoms = h->t.n_atoms;
    if (has_box) *has_box = h->t.has_box ? 1 : 0;
    if (has_time) *has_time = h->t.has_time ? 1 : 0;
    if (format) *format = h->t.format;
    return MDX_OK;
}

int mdx_traj_read_positions(mdx_traj_t h, const int64_t *frames, int64_t n, float *out)
{
    MDX_REQUIRE(h, "NULL handle");
    return h->t.read_positions(frames, n, out);
}

int mdx_traj_read_boxes(mdx_traj_t h, const int64_t *frames, int64_t n, float *boxes6)
{
    MDX_REQUIRE(h, "NULL handle");
    return h->t.read_boxes(frames, n, boxes6);
}

int mdx_traj_read_times(mdx_traj_t h, const int64_t *frames, int64_t n, double *times)
{
    MDX_REQUIRE(h, "NULL handle");
    return h->t.read_times(frames, n, times);
}

int mdx_traj_load_device(mdx_traj_t h, int dev, const int64_t *frames, int64_t n,
                         const int32_t *d_index, int64_t n_sel, float *d_out)
{
    MDX_REQUIRE(h, "NULL handle");
    MDX_REQUIRE(n == 0 || d_out, "NULL output");
    MDX_TRY(set_device(dev));
    if (!h->stream)
        MDX_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    TrajSelection s{d_index, d_index ? n_sel : (n_sel > 0 ? n_sel : h->t.n_atoms), d_out};
    MDX_TRY(h->t.stage_async(dev, h->stream, frames, n, &s, 1));
    MDX_HIP(hipStreamSynchronize(h->stream));
    return MDX_OK;
}

}  // extern "C"
