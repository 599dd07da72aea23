// mdx_traj.hpp — native trajectory ingest, internal C++ view.
//
// Reads the two on-disk layouts the reference's users feed the analysis classes with:
//
//   * AMBER NetCDF trajectories (classic CDF-1 / 64-bit-offset CDF-2 containers — what the
//     reference's own writer produces, /root/reference/src/mdhelper/openmm/file.py:49-52,
//     variables `coordinates`, `cell_lengths`, `cell_angles`, `time` :160-188): big-endian
//     float32 coordinates, one record per frame;
//   * CHARMM/NAMD DCD trajectories (what MDAnalysis hands the reference's tests): Fortran
//     records, x[N] y[N] z[N] planes per frame, optional unit-cell record.
//
// The header is parsed on the host; frames travel raw (file byte order and plane layout)
// through pinned staging buffers to HBM, and one device kernel byte-swaps / transposes /
// gathers them into the float32 [frame][particle][xyz] blocks the analysis kernels read.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "mdx_common.hpp"

namespace mdx {

enum TrajFormat { TRAJ_NETCDF = 1, TRAJ_DCD = 2 };

// One selection of particles to materialise on the device: out[frame][s][xyz] for s < n_sel,
// d_index == nullptr meaning the first n_sel particles of the file.
struct TrajSelection {
    const int *d_index;
    int64_t n_sel;
    float *d_out;
};

struct Trajectory {
    int fd = -1;
    int format = 0;
    int64_t file_bytes = 0;
    int64_t n_frames = 0, n_atoms = 0;
    bool swap = false;          // file byte order differs from the host's
    bool has_box = false, has_time = false;

    // per-frame addressing: byte offset of frame f = first + f * stride
    int64_t frame_stride = 0;
    int64_t coord_first = 0;    // NetCDF: coordinates[f]; DCD: the x record's payload
    int64_t plane_stride = 0;   // DCD: distance between the x, y and z payloads; NetCDF: 0
    int64_t cell_first = -1;    // NetCDF: cell_lengths[f]; DCD: unit-cell payload (6 doubles)
    int64_t angle_first = -1;   // NetCDF: cell_angles[f]
    int64_t time_first = -1;    // NetCDF: time[f]
    int cell_type = 6, angle_type = 6, time_type = 5;   // nc_type (5 float, 6 double)
    double dcd_delta = 0.0;     // DCD: time step between saved frames (AKMA units as stored)
    int64_t dcd_istart = 0, dcd_nsavc = 1;
    double coord_scale = 1.0;   // NetCDF scale_factor attribute of `coordinates`

    // device pipeline (created on first use): frames travel through the device's shared pinned ring
    // (device_stager: its stream, its events — nothing is ever recorded on a caller's stream, so
    // engines may come and go around one open trajectory); d_raw holds one chunk of raw frames
    // between the copy and the unpack kernel of stage_async
    int dev = -1;
    DeviceBuffer d_raw;

    int open(const char *path);
    void close();

    // host reads (native byte order)
    int read_positions(const int64_t *frames, int64_t n, float *out) const;
    int read_boxes(const int64_t *frames, int64_t n, float *boxes6) const;
    int read_times(const int64_t *frames, int64_t n, double *times) const;

    // Raw frames -> pinned -> HBM -> unpack into every selection, on the trajectory's own stream:
    // that stream first waits for what `consumer` holds so far (the outputs may still be read),
    // and `consumer` waits for the unpacked frames.  Returns once the last pinned buffer has been
    // filled; copies and kernels may still be in flight, also across calls.
    int stage_async(int device, hipStream_t consumer, const int64_t *frames, int64_t n,
                    const TrajSelection *sel, int n_sel);

    // The same in two halves, for consumers whose own kernels leave no room on the chip (the RDF's
    // persistent pair kernel holds every wave slot until its launch ends, so a kernel on another
    // stream waits for it — and with it would every copy queued behind that kernel):
    //   stage_raw_async  raw frames -> pinned -> d_raw_out[n][12 n_atoms bytes], copies only (DMA), on
    //                    the ring's stream; `consumer` waits for them;
    //   unpack_async     the unpack kernel(s) on the CALLER's stream, in order with its other kernels.
    int stage_raw_async(int device, hipStream_t consumer, const int64_t *frames, int64_t n, void *d_raw_out);
    int unpack_async(hipStream_t stream, const void *d_raw_in, int64_t n, const TrajSelection *sel,
                     int n_sel) const;

private:
    int parse_netcdf(const std::vector<uint8_t> &head, bool &need_more);
    int parse_dcd(const std::vector<uint8_t> &head, bool &need_more);
    int read_at(int64_t offset, void *dst, size_t bytes) const;
    int fill_raw(const int64_t *frames, int64_t n, uint8_t *dst, HostWorkers *workers = nullptr) const;
    int read_box(int64_t frame, float *box6) const;
};

}  // namespace mdx
