// Compile/link check of the C++ host mirror against libopmgpu.so; also exercises the "no GPU" error path.
#include <cstdio>
#include "opmgpu.hpp"
int main()
{
    std::printf("%s, devices: %d\n", opmgpu_version(), opmgpu_device_count());
    try {
        opmgpu::NewtonIterationBlackoilGpu solver;
        std::printf("solver context created\n");
    } catch (const std::exception& e) {
        std::printf("expected without a GPU: %s\n", e.what());
    }
    return 0;
}
