// corun.hip -- can a VALU-bound role and a store-bound role share a CU?  (feasibility probe for K2's float64 stores)
//
// K2's plane loop runs at the VALU issue bound (19.7 ms at N = 100k without its stores) and its 80 GB of float64 stores need
// >= 12.7 ms of the CU store paths; back to back in one wave they cost 25.6 ms.  This probe runs a register-only stand-in of the
// plane loop (same instruction mix: 768 v_bitop3 + 96 half-rate popcount/pack instructions per stage, 16 stages and barriers per
// "tile", 128 VGPRs, 40 KiB LDS -> 4 workgroups per CU) and a streaming-store role (256 KiB per "tile", 16-byte nontemporal
// stores) as ONE persistent grid whose workgroups pick a role per CU (hardware id + a ticket), and times:
//   0 compute only, 4 workgroups/CU     1 store only, 4/CU        2 roles 3 + 1 per CU      3 roles 2 + 2
//   4 roles 3 + 1, work-conserving (a role whose queue is empty takes the other queue)
//   5 compute only, 3/CU                6 store only, 1/CU
//   7 every workgroup computes and issues the previous tile's 64 stores inside the stage loop, 4 per stage
//   8 every workgroup computes a tile, then issues its 64 stores (what k_mh_compare_p12 does)
//   9 roles 3 + 1 with the store role at wave priority 3 (the compute role runs at 2)
//   hipcc --offload-arch=gfx950 -O3 -o corun corun.hip && ./corun [tiles]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int CHUNK = 256 * 1024;

struct Ctl {
  unsigned next_compute, next_store, pad[2];
  unsigned cu_count[2048];
};

__device__ __forceinline__ unsigned cu_key() {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  return ((xcc & 7u) << 8) | ((hw >> 8) & 0xffu);   // XCD, then se_id / sh_id / cu_id
}

// one compute tile; stores_mode 0: none, 1: 4 of the 64 stores per stage (to st_ptr + q * 4096)
__device__ __forceinline__ void compute_tile(int stores_mode, char *st_ptr) {
  unsigned lo = (unsigned)(uintptr_t)st_ptr, hi = (unsigned)((uintptr_t)st_ptr >> 32);
  register unsigned r114 asm("v114") = lo;
  register unsigned r115 asm("v115") = hi;
  unsigned sm = __builtin_amdgcn_readfirstlane((unsigned)stores_mode);
  asm volatile(
      "s_mov_b32 s40, 16\n\t"
      "v_mov_b32 v64, v114\n\t v_mov_b32 v65, v115\n\t v_mov_b32 v66, v114\n\t v_mov_b32 v67, v115\n\t"
      "v_mov_b32 v68, v114\n\t v_mov_b32 v69, v115\n\t v_mov_b32 v70, v114\n\t v_mov_b32 v71, v115\n\t"
      "v_mov_b32 v72, v114\n\t v_mov_b32 v73, v115\n\t v_mov_b32 v74, v114\n\t v_mov_b32 v75, v115\n\t"
      "v_mov_b32 v76, v114\n\t v_mov_b32 v77, v115\n\t v_mov_b32 v78, v114\n\t v_mov_b32 v79, v115\n\t"
      "s_setprio 2\n\t"
      "1:\n\t"
      "s_barrier\n\t"
      "s_cmp_eq_u32 %2, 0\n\t"
      "s_cbranch_scc1 2f\n\t"
      "global_store_dwordx4 v[114:115], v[116:119], off nt\n\t"
      "v_add_co_u32 v114, vcc, 0x1000, v114\n\t"
      "v_addc_co_u32 v115, vcc, 0, v115, vcc\n\t"
      "global_store_dwordx4 v[114:115], v[116:119], off nt\n\t"
      "v_add_co_u32 v114, vcc, 0x1000, v114\n\t"
      "v_addc_co_u32 v115, vcc, 0, v115, vcc\n\t"
      "global_store_dwordx4 v[114:115], v[116:119], off nt\n\t"
      "v_add_co_u32 v114, vcc, 0x1000, v114\n\t"
      "v_addc_co_u32 v115, vcc, 0, v115, vcc\n\t"
      "global_store_dwordx4 v[114:115], v[116:119], off nt\n\t"
      "v_add_co_u32 v114, vcc, 0x1000, v114\n\t"
      "v_addc_co_u32 v115, vcc, 0, v115, vcc\n\t"
      "2:\n\t"
#include "corun_loop.inc"
      "s_sub_u32 s40, s40, 1\n\t"
      "s_cmp_lg_u32 s40, 0\n\t"
      "s_cbranch_scc1 1b\n\t"
      "s_setprio 0\n\t"
      : "+v"(r114), "+v"(r115)
      : "s"(sm)
      : "memory", "vcc", "scc", "s40", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15",
        "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34",
        "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53",
        "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72",
        "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91",
        "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108",
        "v109", "v110", "v111", "v112", "v113", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
}

__device__ __forceinline__ void store_chunk(char *base) {   // 256 threads x 64 x 16 B
  d2 v = {1.0, 2.0};
  d2 *p = reinterpret_cast<d2 *>(base) + threadIdx.x;
#pragma unroll 16
  for (int q = 0; q < 64; ++q) __builtin_nontemporal_store(v, p + q * 256);
}

__global__ __launch_bounds__(256, 4) void k_corun(Ctl *ctl, char *buf, unsigned buf_chunks, unsigned n_compute, unsigned n_store, int mode) {
  extern __shared__ unsigned char pad_lds[];          // 40 KiB at launch = K2's LDS footprint: 4 workgroups per CU
  __shared__ unsigned s_val[2];
  if (threadIdx.x == 0) {
    s_val[0] = atomicAdd(&ctl->cu_count[cu_key()], 1u);
  }
  __syncthreads();
  const unsigned ticket = s_val[0];
  int role;   // 0 compute, 1 store, 2 exit
  switch (mode) {
    case 0: role = 0; break;
    case 1: role = 1; break;
    case 2: case 4: case 9: role = (ticket & 3u) == 3u ? 1 : 0; break;
    case 3: role = (ticket & 1u) ? 1 : 0; break;
    case 5: role = (ticket & 3u) == 3u ? 2 : 0; break;
    case 6: role = (ticket & 3u) == 0u ? 1 : 2; break;
    default: role = 0; break;
  }
  if (role == 2) return;
  auto take = [&](unsigned *ctr) -> unsigned {
    __syncthreads();
    if (threadIdx.x == 0) s_val[1] = atomicAdd(ctr, 1u);
    __syncthreads();
    return s_val[1];
  };
  if (mode == 7 || mode == 8) {
    unsigned prev = 0xffffffffu;
    for (;;) {
      const unsigned t = take(&ctl->next_compute);
      if (t >= n_compute) break;
      if (mode == 7) {
        compute_tile(prev != 0xffffffffu ? 1 : 0, buf + (size_t)((prev == 0xffffffffu ? t : prev) % buf_chunks) * CHUNK + threadIdx.x * 16);
      } else {
        compute_tile(0, buf);
        store_chunk(buf + (size_t)(t % buf_chunks) * CHUNK);
      }
      prev = t;
    }
    if (mode == 7 && prev != 0xffffffffu) store_chunk(buf + (size_t)(prev % buf_chunks) * CHUNK);
    return;
  }
  for (int pass = 0; pass < 2; ++pass) {
    if (role == 0) {
      for (;;) {
        const unsigned t = take(&ctl->next_compute);
        if (t >= n_compute) break;
        compute_tile(0, buf);
      }
    } else {
      if (mode == 9) __builtin_amdgcn_s_setprio(3);   // the store role issues few instructions: let it go first
      for (;;) {
        const unsigned t = take(&ctl->next_store);
        if (t >= n_store) break;
        store_chunk(buf + (size_t)(t % buf_chunks) * CHUNK);
      }
    }
    if (mode != 4) break;
    role ^= 1;   // work-conserving: help the other queue
  }
}

int main(int argc, char **argv) {
  const unsigned tiles = argc > 1 ? (unsigned)atoi(argv[1]) : 306153u;   // K2 at N = 100k: 782 * 783 / 2 tiles, 256 KiB of float64 each
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const unsigned buf_chunks = 65536;   // 16 GiB
  char *buf;
  Ctl *ctl;
  CHECK(hipMalloc(&buf, (size_t)buf_chunks * CHUNK));
  CHECK(hipMalloc(&ctl, sizeof(Ctl)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const double lane_ops = (double)tiles * 16 * 768 * 256, bytes = (double)tiles * CHUNK;
  printf("%d CUs, %u tiles: %.3g lane-v_bitop3 (+ 1/8 as many half-rate count ops), %.1f GB of stores\n", cus, tiles, lane_ops, bytes / 1e9);
  const char *names[] = {"compute only 4/CU", "store only 4/CU", "roles 3+1", "roles 2+2", "roles 3+1 work-conserving", "compute only 3/CU",
                         "store only 1/CU", "in-loop stores (4 per stage)", "compute then 64 stores", "roles 3+1, store role at priority 3"};
  for (int mode = 0; mode <= 9; ++mode) {
    float best = 1e30f;
    std::vector<unsigned> hist(8, 0);
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipMemset(ctl, 0, sizeof(Ctl)));
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_corun, dim3(cus * 4), dim3(256), 40 * 1024 - 64, 0, ctl, buf, buf_chunks, tiles, tiles, mode);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
      if (rep == 0) {
        Ctl h;
        CHECK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
        for (int i = 0; i < 2048; ++i) if (h.cu_count[i]) hist[h.cu_count[i] < 7 ? h.cu_count[i] : 7]++;
      }
    }
    const bool c = mode != 1 && mode != 6, s = mode != 0 && mode != 5;
    printf("mode %d %-30s %8.2f ms", mode, names[mode], best);
    if (c) printf("  compute %.1f T lane-bitop3/s", lane_ops / best / 1e9);
    if (s) printf("  stores %.2f TB/s", bytes / best / 1e9);
    printf("   [workgroups per CU key: 1:%u 2:%u 3:%u 4:%u 5:%u 6+:%u]\n", hist[1], hist[2], hist[3], hist[4], hist[5], hist[6] + hist[7]);
  }
  return 0;
}
