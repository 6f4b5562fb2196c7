// ply.hip -- snapshot format (Data/PlyWriter.swift:22-113 writer, :149-233 loader)
//
// The reference walks the six tensors on the host one float at a time.  Here the interleave is one HBM-rate kernel
// ([N, 14 + 3M] rows, the file's vertex layout), the rows cross PCIe once through pinned memory, and the host only
// writes the header and the blob.  Loading is the mirror image, with the header's property order honoured.
#include <errno.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>

#include <string>
#include <vector>

#include "gs_ctx.h"

namespace gs {

constexpr int PLY_THREADS = 256;

// vertex row: x y z | f_dc_0..2 | f_rest_0..3M-1 (coefficient-major, channel-minor: PlyWriter.swift:139-140 flattens
// [N, M, 3] as is) | opacity | scale_0..2 | rot_0..3
__global__ __launch_bounds__(PLY_THREADS) void ply_pack_kernel(long long total, int F, int L,
                                                               const float* __restrict__ xyz,
                                                               const float* __restrict__ fdc,
                                                               const float* __restrict__ frest,
                                                               const float* __restrict__ opacity,
                                                               const float* __restrict__ scales,
                                                               const float* __restrict__ rot, float* __restrict__ rows)
{
    const long long e = (long long)blockIdx.x * PLY_THREADS + threadIdx.x;
    if (e >= total) return;
    const long long j = e / F;
    int f = (int)(e - j * F);
    float v;
    if (f < 3) v = xyz[j * 3 + f];
    else if ((f -= 3) < 3) v = fdc[j * 3 + f];
    else if ((f -= 3) < L) v = frest[j * L + f];
    else if ((f -= L) < 1) v = opacity[j];
    else if ((f -= 1) < 3) v = scales[j * 3 + f];
    else v = rot[j * 4 + (f - 3)];
    rows[e] = v;
}

struct PlyFieldMap {
    int xyz[3], fdc[3], opacity, scales[3], rot[4];
};

// one thread per OUTPUT float; fieldRest[L] (device) gives the row column of each f_rest_i
__global__ __launch_bounds__(PLY_THREADS) void ply_unpack_kernel(long long total, int F, int L, int stride,
                                                                 PlyFieldMap map, const int* __restrict__ fieldRest,
                                                                 const float* __restrict__ rows,
                                                                 float* __restrict__ xyz, float* __restrict__ fdc,
                                                                 float* __restrict__ frest, float* __restrict__ opacity,
                                                                 float* __restrict__ scales, float* __restrict__ rot)
{
    const long long e = (long long)blockIdx.x * PLY_THREADS + threadIdx.x;
    if (e >= total) return;
    const long long j = e / F;
    int f = (int)(e - j * F);
    const float* row = rows + j * stride;
    if (f < 3) xyz[j * 3 + f] = row[map.xyz[f]];
    else if ((f -= 3) < 3) fdc[j * 3 + f] = row[map.fdc[f]];
    else if ((f -= 3) < L) frest[j * L + f] = row[fieldRest[f]];
    else if ((f -= L) < 1) opacity[j] = row[map.opacity];
    else if ((f -= 1) < 3) scales[j * 3 + f] = row[map.scales[f]];
    else rot[j * 4 + (f - 3)] = row[map.rot[f - 3]];
}

int launch_ply_pack(gs_ctx* c, int N, int K, const float* xyz, const float* fdc, const float* frest,
                    const float* opacity, const float* scales, const float* rot, float* rows)
{
    if (N == 0) return GS_OK;
    const int L = (K - 1) * 3, F = 14 + L;
    const long long total = (long long)N * F;
    hipLaunchKernelGGL(ply_pack_kernel, dim3(gs_div_up(total, PLY_THREADS)), dim3(PLY_THREADS), 0, c->stream, total, F, L,
                       xyz, fdc, frest, opacity, scales, rot, rows);
    GS_HIP_CHECK(c, hipGetLastError());
    return GS_OK;
}

// ---- header ------------------------------------------------------------------------------------------------
struct PlyHeader {
    long long numPoints = 0;
    int M = -1, D = -1;
    std::vector<std::string> fields;
    size_t dataOffset = 0;
};

static std::vector<std::string> split_ws(const std::string& line)
{
    std::vector<std::string> parts;
    size_t i = 0;
    while (i < line.size()) {
        while (i < line.size() && line[i] == ' ') i++;
        size_t j = i;
        while (j < line.size() && line[j] != ' ') j++;
        if (j > i) parts.push_back(line.substr(i, j - i));
        i = j;
    }
    return parts;
}

// PlyWriter.swift:149-183.  Only `property float <name>` lines count as fields; anything else is skipped.
static int parse_ply_header(gs_ctx* c, FILE* fp, PlyHeader& h)
{
    std::string head;
    const std::string marker = "end_header\n";
    char buf[4096];
    bool found = false;
    while (!found) {
        const size_t got = fread(buf, 1, sizeof(buf), fp);
        if (got == 0) break;
        head.append(buf, got);
        const size_t pos = head.find(marker);
        if (pos != std::string::npos) { h.dataOffset = pos + marker.size(); found = true; }
        if (head.size() > (1u << 24)) break;
    }
    if (!found) { c->err = "PLY: no end_header"; return GS_ERR_IO; }
    head.resize(h.dataOffset);
    size_t p = 0;
    while (p < head.size()) {
        size_t q = head.find('\n', p);
        if (q == std::string::npos) q = head.size();
        const std::string line = head.substr(p, q - p);
        p = q + 1;
        const std::vector<std::string> parts = split_ws(line);
        if (line.rfind("comment features_rest_shape", 0) == 0 && parts.size() >= 4) {
            h.M = atoi(parts[2].c_str());
            h.D = atoi(parts[3].c_str());
        }
        if (parts.size() >= 3 && parts[0] == "element" && parts[1] == "vertex") h.numPoints = atoll(parts[2].c_str());
        else if (parts.size() == 3 && parts[0] == "property" && parts[1] == "float") h.fields.push_back(parts[2]);
    }
    if (h.M < 0 || h.D < 0) { c->err = "PLY: no features_rest_shape comment"; return GS_ERR_IO; }
    return GS_OK;
}

static int field_index(const PlyHeader& h, const std::string& name)
{
    for (size_t i = 0; i < h.fields.size(); i++)
        if (h.fields[i] == name) return (int)i;
    return -1;
}

static void make_parent_dirs(const std::string& path)
{
    for (size_t i = 1; i < path.size(); i++)
        if (path[i] == '/') (void)mkdir(path.substr(0, i).c_str(), 0777);
}

int ply_write_file(gs_ctx* c, const char* path, int N, int K, const float* xyz, const float* fdc, const float* frest,
                   const float* opacity, const float* scales, const float* rot)
{
    const int M = K - 1, F = 14 + 3 * M;
    const size_t bytes = sizeof(float) * (size_t)N * F;
    float* rowsDev = nullptr;
    float* rowsHost = nullptr;
    int rc = GS_OK;
    if (N > 0) {
        GS_HIP_CHECK(c, hipMalloc(&rowsDev, bytes));
        if (hipHostMalloc(&rowsHost, bytes, hipHostMallocDefault) != hipSuccess) {
            (void)hipFree(rowsDev);
            c->err = "PLY: pinned staging allocation failed";
            return GS_ERR_HIP;
        }
        rc = launch_ply_pack(c, N, K, xyz, fdc, frest, opacity, scales, rot, rowsDev);
        if (rc == GS_OK && (hipMemcpyAsync(rowsHost, rowsDev, bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                            hipStreamSynchronize(c->stream) != hipSuccess)) {
            c->err = "PLY: device to host copy failed";
            rc = GS_ERR_HIP;
        }
    }
    if (rc == GS_OK) {
        std::string header = "ply\nformat binary_little_endian 1.0\n";
        header += "comment features_rest_shape " + std::to_string(M) + " 3\n";
        header += "element vertex " + std::to_string(N) + "\n";
        header += "property float x\nproperty float y\nproperty float z\n";
        header += "property float f_dc_0\nproperty float f_dc_1\nproperty float f_dc_2\n";
        for (int i = 0; i < M * 3; i++) header += "property float f_rest_" + std::to_string(i) + "\n";
        header += "property float opacity\nproperty float scale_0\nproperty float scale_1\nproperty float scale_2\n";
        header += "property float rot_0\nproperty float rot_1\nproperty float rot_2\nproperty float rot_3\nend_header\n";
        make_parent_dirs(path);                               // PlyWriter.swift:106-111
        FILE* fp = fopen(path, "wb");
        if (!fp) { c->err = std::string("PLY: cannot open for writing: ") + strerror(errno); rc = GS_ERR_IO; }
        else {
            bool ok = fwrite(header.data(), 1, header.size(), fp) == header.size();
            if (ok && bytes) ok = fwrite(rowsHost, 1, bytes, fp) == bytes;      // gfx950 hosts are little-endian
            if (fclose(fp) != 0) ok = false;
            if (!ok) { c->err = "PLY: short write"; rc = GS_ERR_IO; }
        }
    }
    if (rowsHost) (void)hipHostFree(rowsHost);
    if (rowsDev) (void)hipFree(rowsDev);
    return rc;
}

int ply_probe_file(gs_ctx* c, const char* path, long long* N, int* M, int* D)
{
    FILE* fp = fopen(path, "rb");
    if (!fp) { c->err = std::string("PLY: cannot open: ") + strerror(errno); return GS_ERR_IO; }
    PlyHeader h;
    const int rc = parse_ply_header(c, fp, h);
    fclose(fp);
    if (rc) return rc;
    *N = h.numPoints; *M = h.M; *D = h.D;
    return GS_OK;
}

int ply_load_file(gs_ctx* c, const char* path, int N, int K, float* xyz, float* fdc, float* frest, float* opacity,
                  float* scales, float* rot)
{
    FILE* fp = fopen(path, "rb");
    if (!fp) { c->err = std::string("PLY: cannot open: ") + strerror(errno); return GS_ERR_IO; }
    PlyHeader h;
    int rc = parse_ply_header(c, fp, h);
    if (rc) { fclose(fp); return rc; }
    const int M = K - 1, L = 3 * M;
    if (h.numPoints != N || h.M != M || h.D != 3) {
        fclose(fp);
        c->err = "PLY: N / features_rest_shape differ from the caller's buffers (probe first; D must be 3)";
        return GS_ERR_SIZE_MISMATCH;
    }
    PlyFieldMap map;
    std::vector<int> rest(L > 0 ? L : 1, 0);
    bool ok = true;
    const char* names3[3] = {"x", "y", "z"};
    for (int i = 0; i < 3; i++) ok &= (map.xyz[i] = field_index(h, names3[i])) >= 0;
    for (int i = 0; i < 3; i++) ok &= (map.fdc[i] = field_index(h, "f_dc_" + std::to_string(i))) >= 0;
    for (int i = 0; i < L; i++) ok &= (rest[i] = field_index(h, "f_rest_" + std::to_string(i))) >= 0;
    ok &= (map.opacity = field_index(h, "opacity")) >= 0;
    for (int i = 0; i < 3; i++) ok &= (map.scales[i] = field_index(h, "scale_" + std::to_string(i))) >= 0;
    for (int i = 0; i < 4; i++) ok &= (map.rot[i] = field_index(h, "rot_" + std::to_string(i))) >= 0;
    if (!ok) { fclose(fp); c->err = "PLY: a required float property is missing"; return GS_ERR_IO; }
    if (N == 0) { fclose(fp); return GS_OK; }
    const int stride = (int)h.fields.size();
    const size_t bytes = sizeof(float) * (size_t)N * stride;
    float* rowsHost = nullptr;
    float* rowsDev = nullptr;
    int* restDev = nullptr;
    if (hipHostMalloc(&rowsHost, bytes, hipHostMallocDefault) != hipSuccess) {
        fclose(fp);
        c->err = "PLY: pinned staging allocation failed";
        return GS_ERR_HIP;
    }
    if (fseek(fp, (long)h.dataOffset, SEEK_SET) != 0 || fread(rowsHost, 1, bytes, fp) != bytes) {
        c->err = "PLY: vertex data shorter than the header says";
        rc = GS_ERR_IO;
    }
    fclose(fp);
    if (rc == GS_OK && (hipMalloc(&rowsDev, bytes) != hipSuccess || hipMalloc(&restDev, sizeof(int) * rest.size()) != hipSuccess)) {
        c->err = "PLY: device staging allocation failed";
        rc = GS_ERR_HIP;
    }
    if (rc == GS_OK) {
        const int F = 14 + L;
        const long long total = (long long)N * F;
        if (hipMemcpyAsync(rowsDev, rowsHost, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipMemcpyAsync(restDev, rest.data(), sizeof(int) * rest.size(), hipMemcpyHostToDevice, c->stream) != hipSuccess) {
            c->err = "PLY: host to device copy failed";
            rc = GS_ERR_HIP;
        } else {
            hipLaunchKernelGGL(ply_unpack_kernel, dim3(gs_div_up(total, PLY_THREADS)), dim3(PLY_THREADS), 0, c->stream,
                               total, F, L, stride, map, restDev, rowsDev, xyz, fdc, frest, opacity, scales, rot);
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
                c->err = "PLY: unpack kernel failed";
                rc = GS_ERR_HIP;
            }
        }
    }
    if (restDev) (void)hipFree(restDev);
    if (rowsDev) (void)hipFree(rowsDev);
    (void)hipHostFree(rowsHost);
    return rc;
}

}  // namespace gs
