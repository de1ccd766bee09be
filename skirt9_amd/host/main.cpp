// main.cpp -- command line driver: runs an unchanged SKIRT ski file with the primary-emission loop on the MI355X.
//
//   skirt_mi355x [-o outdir] [-g device] [-n packets] [-s] file.ski     (-s: particle-medium densities sampled on the GPU)
//
// Counterpart of SKIRT/main (SkirtCommandLineHandler.cpp:295-372 doSimulation): construct the simulation from the
// ski file, set it up, run the primary emission segment (here: on the GPU through the C ABI of include/pmc.h),
// write the instrument output.  There is no CPU fallback: without a HIP device pmc_create fails and so does the run.

#include "../../include/pmc.h"
#include "../../include/skirt_host.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

int main(int argc, char** argv)
{
    std::string outdir = ".", ski;
    int device = 0;
    unsigned long long packets = 0;
    bool deviceSetup = false;
    for (int i = 1; i < argc; ++i)
    {
        if (!strcmp(argv[i], "-s"))
        {
            deviceSetup = true;  // sample the densities of an imported particle medium on the GPU during setup
            continue;
        }
        else if (!strcmp(argv[i], "-o") && i + 1 < argc)
            outdir = argv[++i];
        else if (!strcmp(argv[i], "-g") && i + 1 < argc)
            device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-n") && i + 1 < argc)
            packets = strtoull(argv[++i], nullptr, 10);
        else
            ski = argv[i];
    }
    if (ski.empty())
    {
        fprintf(stderr, "usage: skirt_mi355x [-o outdir] [-g device] [-n packets] [-s] file.ski\n");
        return 2;
    }
    using clock = std::chrono::steady_clock;
    auto seconds = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };

    skh_simulation* sim = skh_load(ski.c_str());
    if (!sim)
    {
        fprintf(stderr, "Fatal error: %s\n", skh_last_error());
        return 1;
    }
    if (packets) skh_set_num_packets(sim, packets);
    if (deviceSetup
        && skh_set_particle_sampler(sim, reinterpret_cast<void*>(&pmc_sampler_create), reinterpret_cast<void*>(&pmc_sampler_density),
                                    reinterpret_cast<void*>(&pmc_sampler_destroy), reinterpret_cast<void*>(&pmc_last_error), device)
               != 0)
    {
        fprintf(stderr, "Fatal error: %s\n", skh_last_error());
        return 1;
    }
    auto t0 = clock::now();
    printf("Starting setup...\n");
    if (skh_setup(sim) != 0)
    {
        fprintf(stderr, "Fatal error: %s\n", skh_last_error());
        return 1;
    }
    char summary[2048];
    skh_summary(sim, summary, sizeof(summary));
    auto t1 = clock::now();
    printf("%sFinished setup in %.1f s.\n", summary, seconds(t0, t1));

    pmc_ctx* ctx = nullptr;
    if (pmc_create(skh_scene(sim), device, &ctx) != PMC_OK)
    {
        fprintf(stderr, "Fatal error: %s\n", pmc_last_error());
        return 1;
    }
    const unsigned long long n = skh_num_packets(sim);
    printf("Launching %g primary emission photon packets\n", (double)n);
    auto t2 = clock::now();
    std::vector<double> frames(skh_frame_size(sim));
    if (pmc_run_primary(ctx, 0, n, (uint64_t)skh_seed(sim)) != PMC_OK || pmc_download(ctx, frames.data(), (int64_t)frames.size()) != PMC_OK)
    {
        fprintf(stderr, "Fatal error: %s\n", pmc_last_error());
        return 1;
    }
    auto t3 = clock::now();
    pmc_counter_values c;
    pmc_counters(ctx, &c);
    printf("Finished primary emission in %.3f s (%.3g packets/s; %.1f cell visits and %.1f detector updates per packet).\n",
           seconds(t2, t3), n / seconds(t2, t3), (double)c.cell_visits / n, (double)c.detector_updates / n);
    if (skh_write(sim, frames.data(), outdir.c_str()) != 0)
    {
        fprintf(stderr, "Fatal error: %s\n", skh_last_error());
        return 1;
    }
    // the radiation field, if the ski file stores it: downloaded after the segment and written by the configured
    // RadiationFieldProbe (the reference sums it over processes first, MediumSystem.cpp:1304-1313; one GPU here)
    if (const int64_t rfSize = skh_radiation_field_size(sim))
    {
        std::vector<double> rf(rfSize);
        if (pmc_download_radiation_field(ctx, rf.data(), rfSize) != PMC_OK)
        {
            fprintf(stderr, "Fatal error: %s\n", pmc_last_error());
            return 1;
        }
        if (skh_write_radiation_field(sim, rf.data(), outdir.c_str()) != 0)
        {
            fprintf(stderr, "Fatal error: %s\n", skh_last_error());
            return 1;
        }
    }
    printf("Finished final output in %.1f s.\n", seconds(t3, clock::now()));
    pmc_destroy(ctx);
    skh_free(sim);
    return 0;
}
