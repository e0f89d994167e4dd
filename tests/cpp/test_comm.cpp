// The C path of the config-5 exchange (include/hyslam_amd.h: hs_comm_*; hyslam_amd/csrc/hs_comm.hip) without Python or torch:
// one process per rank, the unique id travels from rank 0 to the others through a file, every rank creates its communicator on its GPU,
// gathers one record per rank (out of place and in place) on the extractor handle's stream and checks every peer's bytes.
// usage: test_comm ID_FILE WORLD RANK [RECORD_BYTES]
//   no GPU / no librccl: every rank prints "NO DEVICE" and exits 0 before ncclCommInitRank (a CPU-only box still builds and starts the ranks);
//   fewer GPUs than ranks: "NOT ENOUGH DEVICES" and exit 0 (RCCL wants one device per rank).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <hip/hip_runtime_api.h>
#include "../../include/hyslam_amd.h"

#define FAIL(code, ...) do { printf("rank %d: ", rank); printf(__VA_ARGS__); printf("\n"); return code; } while (0)

int main(int argc, char** argv)
{
    if (argc < 4) { printf("usage: test_comm ID_FILE WORLD RANK [RECORD_BYTES]\n"); return 2; }
    const std::string id_file = argv[1];
    const int world = atoi(argv[2]), rank = atoi(argv[3]);
    const size_t rb = argc > 4 ? (size_t)atoll(argv[4]) : hs_record_bytes(2012);
    int ndev = 0;
    if (hs_device_count(&ndev) != HS_OK || ndev == 0) {
        uint8_t id[HS_COMM_ID_BYTES];
        if (hs_comm_get_unique_id(id) == HS_OK) FAIL(3, "a unique id without a device");
        hs_comm* c = reinterpret_cast<hs_comm*>(1);
        if (hs_comm_create(nullptr, id, world, rank, &c) != HS_ERR_INVALID) FAIL(3, "a null handle was accepted");
        printf("NO DEVICE (rank %d of %d)\n", rank, world);
        return 0;
    }
    if (ndev < world) { printf("NOT ENOUGH DEVICES (%d for %d ranks)\n", ndev, world); return 0; }
    // ---- the id: rank 0 creates it, the others wait for the file
    uint8_t id[HS_COMM_ID_BYTES];
    if (rank == 0) {
        if (hs_comm_get_unique_id(id) != HS_OK) FAIL(4, "hs_comm_get_unique_id failed");
        const std::string tmp = id_file + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id)) FAIL(4, "cannot write %s", tmp.c_str());
        fclose(f);
        if (rename(tmp.c_str(), id_file.c_str()) != 0) FAIL(4, "rename failed");
    } else {
        bool got = false;
        for (int i = 0; i < 600 && !got; i++) {
            FILE* f = fopen(id_file.c_str(), "rb");
            if (f) { got = fread(id, 1, sizeof(id), f) == sizeof(id); fclose(f); }
            if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
        if (!got) FAIL(4, "no id file after 60 s");
    }
    hs_orb_params p; hs_orb_default_params(&p);
    hs_orb* h = nullptr;
    if (hs_orb_create(&p, rank, &h) != HS_OK) FAIL(5, "hs_orb_create on device %d failed", rank);
    hs_comm* c = nullptr;
    if (hs_comm_create(h, id, world, rank, &c) != HS_OK) FAIL(6, "hs_comm_create: %s", hs_orb_last_error(h));
    if (hs_comm_world(c) != world || hs_comm_rank(c) != rank) FAIL(7, "world / rank accessors");
    // ---- one record per rank: byte j of rank r's record = (r * 131 + j * 7 + (j >> 8)) & 255
    auto pattern = [](int r, size_t j) { return (uint8_t)((r * 131 + j * 7 + (j >> 8)) & 255); };
    std::vector<uint8_t> mine(rb), all(rb * world);
    for (size_t j = 0; j < rb; j++) mine[j] = pattern(rank, j);
    uint8_t *d_rec = nullptr, *d_all = nullptr;
    if (hipSetDevice(rank) != hipSuccess || hipMalloc((void**)&d_rec, rb) != hipSuccess || hipMalloc((void**)&d_all, rb * world) != hipSuccess) FAIL(8, "hipMalloc");
    for (int mode = 0; mode < 2; mode++) {                                  // 0: out of place, 1: in place (record already at its slot)
        if (hipMemset(d_all, 0xEE, rb * world) != hipSuccess) FAIL(8, "hipMemset");
        const uint8_t* src = d_rec;
        if (mode == 1) src = d_all + (size_t)rank * rb;
        if (hipMemcpy((void*)src, mine.data(), rb, hipMemcpyHostToDevice) != hipSuccess) FAIL(8, "hipMemcpy");
        if (hs_comm_allgather_records(c, src, d_all, rb, nullptr) != HS_OK) FAIL(9, "allgather: %s", hs_comm_last_error(c));
        if (hs_orb_synchronize(h, nullptr) != HS_OK) FAIL(9, "synchronize");
        if (hipMemcpy(all.data(), d_all, rb * world, hipMemcpyDeviceToHost) != hipSuccess) FAIL(8, "hipMemcpy back");
        for (int r = 0; r < world; r++)
            for (size_t j = 0; j < rb; j++)
                if (all[(size_t)r * rb + j] != pattern(r, j)) FAIL(10, "mode %d: record of rank %d differs at byte %zu", mode, r, j);
    }
    if (hs_comm_allgather_records(c, nullptr, d_all, rb, nullptr) != HS_ERR_INVALID) FAIL(11, "null record accepted");
    (void)hipFree(d_rec); (void)hipFree(d_all);
    hs_comm_destroy(c);
    hs_orb_destroy(h);
    printf("COMM OK rank %d of %d, %zu-byte records\n", rank, world, rb);
    return 0;
}
