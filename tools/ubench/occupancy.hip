// Micro-benchmark (bring-up evidence, not product): how many 256-thread workgroups that use N VGPRs per
// wave are co-resident per CU on gfx950?  Each workgroup spins ~200 us and records start/end clocks;
// co-residency = number of workgroups that started before the first one ended.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int REG>
__global__ void __launch_bounds__(256, 2) spin(long long *clk, unsigned *ids, long long cycles) {
    if (REG >= 255) asm volatile("v_mov_b32 v255, 0" ::: "v255");
    else if (REG >= 247) asm volatile("v_mov_b32 v247, 0" ::: "v247");
    else if (REG >= 239) asm volatile("v_mov_b32 v239, 0" ::: "v239");
    else if (REG >= 127) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = t0;
        clk[2 * blockIdx.x + 1] = wall_clock64();
        ids[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
}

__global__ void __launch_bounds__(256, 2) spin_scratch(long long *clk, unsigned *ids, long long cycles, int k) {
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    volatile float priv[96];                       // dynamic indexing -> scratch (private segment)
    for (int i = 0; i < 96; ++i) priv[(i * k) % 96] = i;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = t0;
        clk[2 * blockIdx.x + 1] = wall_clock64() + (long long)(priv[k % 96] * 0.f);
        ids[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
}

void run_scratch() {
    const int grid = 2048;
    long long *clk; unsigned *ids;
    (void)hipMalloc(&clk, grid * 16); (void)hipMalloc(&ids, grid * 4);
    hipLaunchKernelGGL(spin_scratch, dim3(grid), dim3(256), 24 * 1024, 0, clk, ids, 20000LL, 7);
    (void)hipDeviceSynchronize();
    std::vector<long long> c(2 * grid);
    (void)hipMemcpy(c.data(), clk, grid * 16, hipMemcpyDeviceToHost);
    long long first_end = c[1];
    for (int i = 0; i < grid; ++i) first_end = std::min(first_end, c[2 * i + 1]);
    int co = 0;
    for (int i = 0; i < grid; ++i) co += c[2 * i] < first_end;
    printf("%-28s : %4d workgroups co-resident (%.2f per CU)\n", "256 vgpr + 384 B scratch + 24 KB LDS", co, co / 256.0);
}

template <int REG>
void run(const char *name, size_t dyn_lds) {
    const int grid = 2048;
    long long *clk; unsigned *ids;
    (void)hipMalloc(&clk, grid * 16); (void)hipMalloc(&ids, grid * 4);
    hipLaunchKernelGGL(spin<REG>, dim3(grid), dim3(256), dyn_lds, 0, clk, ids, 20000LL);   // wall clock is 100 MHz: 200 us
    (void)hipDeviceSynchronize();
    std::vector<long long> c(2 * grid);
    (void)hipMemcpy(c.data(), clk, grid * 16, hipMemcpyDeviceToHost);
    long long first_end = c[1];
    for (int i = 0; i < grid; ++i) first_end = std::min(first_end, c[2 * i + 1]);
    int co = 0;
    for (int i = 0; i < grid; ++i) co += c[2 * i] < first_end;
    printf("%-28s regs>=%3d dynLDS=%6zu : %4d workgroups co-resident (%.2f per CU)\n", name, REG, dyn_lds, co, co / 256.0);
    (void)hipFree(clk); (void)hipFree(ids);
}

int main() {
    run<0>("small", 0);
    run<127>("128 vgpr", 0);
    run<239>("240 vgpr", 0);
    run<247>("248 vgpr", 0);
    run<255>("256 vgpr", 0);
    run<255>("256 vgpr + 24 KB LDS", 24 * 1024);
    run<255>("256 vgpr + 56 KB LDS", 56 * 1024);
    run<239>("240 vgpr + 56 KB LDS", 56 * 1024);
    run_scratch();
    return 0;
}
