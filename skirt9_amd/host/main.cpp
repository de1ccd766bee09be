// main.cpp -- command line driver: runs an unchanged SKIRT ski file with the primary-emission loop on the MI355X.
//
//   skirt_mi355x [-o outdir] [-g device[,device...]] [-n packets] [-s] [--rccl] file.ski
//        -g 0,1,2,3   the devices that share the segment: one host thread per device, the histories split statically by
//                     index (pmc_history_range), ONE RCCL reduce of the detector arrays onto the first device at the end
//                     of the segment (ProcessManager::sumToRoot, FluxRecorder.cpp:487-493), which writes the output
//        -s           particle-medium densities sampled on the GPU during setup
//        --rccl       run the collective also with a single device (a one-rank communicator: checks the call path)
//
// Counterpart of SKIRT/main (SkirtCommandLineHandler.cpp:295-372 doSimulation): construct the simulation from the
// ski file, set it up, run the primary emission segment (here: on the GPU through the C ABI of include/pmc.h),
// write the instrument output.  There is no CPU fallback: without a HIP device pmc_create fails and so does the run.

#include "../../include/pmc.h"
#include "../../include/skirt_host.h"
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

int main(int argc, char** argv)
{
    std::string outdir = ".", ski;
    std::vector<int32_t> devices;
    unsigned long long packets = 0;
    bool deviceSetup = false, forceComm = false;
    for (int i = 1; i < argc; ++i)
    {
        if (!strcmp(argv[i], "-s"))
        {
            deviceSetup = true;  // sample the densities of an imported particle medium on the GPU during setup
            continue;
        }
        else if (!strcmp(argv[i], "-o") && i + 1 < argc)
            outdir = argv[++i];
        else if (!strcmp(argv[i], "--rccl"))
            forceComm = true;
        else if (!strcmp(argv[i], "-g") && i + 1 < argc)
        {
            // a comma-separated list of distinct, non-negative device indices; anything else is a usage error
            const char* p = argv[++i];
            bool ok = *p != 0;
            while (ok && *p)
            {
                char* end = nullptr;
                const long v = strtol(p, &end, 10);
                ok = end != p && v >= 0 && v < 1024 && (*end == ',' || *end == 0) && !(*end == ',' && end[1] == 0);
                for (int32_t d : devices) ok = ok && d != (int32_t)v;
                if (ok) devices.push_back((int32_t)v);
                p = *end == ',' ? end + 1 : end;
            }
            if (!ok)
            {
                fprintf(stderr, "skirt_mi355x: -g takes a comma-separated list of distinct device indices, e.g. -g 0,1,2,3\n");
                return 2;
            }
        }
        else if (!strcmp(argv[i], "-n") && i + 1 < argc)
            packets = strtoull(argv[++i], nullptr, 10);
        else
            ski = argv[i];
    }
    if (ski.empty())
    {
        fprintf(stderr, "usage: skirt_mi355x [-o outdir] [-g device[,device...]] [-n packets] [-s] [--rccl] file.ski\n");
        return 2;
    }
    if (devices.empty()) devices.push_back(0);
    const int device = devices[0];
    const int G = (int)devices.size();
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

    // ---- the segment: one host thread and one engine context per device, every device a replica of the scene and a
    //      static range of the history indices; one RCCL reduce of the detector arrays onto the first device
    const unsigned long long n = skh_num_packets(sim);
    const bool useComm = G > 1 || forceComm;
    std::vector<void*> comms(G, nullptr);
    if (useComm && pmc_comm_init_all(G, devices.data(), comms.data()) != PMC_OK)
    {
        fprintf(stderr, "Fatal error: %s\n", pmc_last_error());
        return 1;
    }
    printf("Launching %g primary emission photon packets on %d device%s%s\n", (double)n, G, G > 1 ? "s" : "",
           useComm ? " (detector arrays summed over RCCL)" : "");
    auto t2 = clock::now();
    std::vector<pmc_ctx*> ctxs(G, nullptr);
    std::vector<std::string> errors(G), statisticsLost(G);
    std::vector<pmc_counter_values> counts(G);
    std::vector<double> frames(skh_frame_size(sim));
    std::vector<double> rf(skh_radiation_field_size(sim));
    // Every device thread either enters the collectives or none does: a thread whose context or segment failed would
    // otherwise leave the others waiting in ncclReduce for ever.  The threads meet once after their segments; if any of
    // them failed, all skip the collectives.
    std::mutex meetLock;
    std::condition_variable meetCv;
    int arrived = 0, failures = 0;
    auto meet = [&](bool ok) {
        std::unique_lock<std::mutex> lock(meetLock);
        arrived += 1;
        if (!ok) failures += 1;
        if (arrived == G)
            meetCv.notify_all();
        else
            meetCv.wait(lock, [&] { return arrived == G; });
        return failures == 0;
    };
    auto work = [&](int g) {
        auto failed = [&]() { errors[g] = pmc_last_error(); };
        bool ok = pmc_create(skh_scene(sim), devices[g], &ctxs[g]) == PMC_OK;
        if (ok)
        {
            // (the first device reports for all: MonteCarloSimulation::logProgress / Log::infoIfElapsed, every 3 s)
            if (g == 0)
                pmc_set_progress(ctxs[g], [](void*, uint64_t launched, uint64_t total) {
                    printf("Launched primary emission photon packets: %.1f%%\n", total ? 100. * (double)launched / (double)total : 100.);
                    fflush(stdout);
                }, nullptr, 3.);
            uint64_t first = 0, count = 0;
            pmc_history_range(n, g, G, &first, &count);
            const int rc = pmc_run_primary(ctxs[g], first, count, (uint64_t)skh_seed(sim));
            // (the pool of statistics list blocks ran out: the flux arrays of the segment are complete, its sums of w^k are not --
            // the run goes on, and the statistics files are left out below)
            if (rc == PMC_ERR_OVERFLOW) statisticsLost[g] = pmc_last_error();
            ok = rc == PMC_OK || rc == PMC_ERR_OVERFLOW;
        }
        if (!ok) failed();
        if (ok) pmc_counters(ctxs[g], &counts[g]);
        if (!meet(ok) || !ok) return;
        if (useComm && pmc_reduce_frames(ctxs[g], comms[g], 0) != PMC_OK) return failed();
        if (useComm && !rf.empty() && pmc_allreduce_radiation_field(ctxs[g], comms[g]) != PMC_OK) return failed();
        if (g == 0)
        {
            if (pmc_download(ctxs[g], frames.data(), (int64_t)frames.size()) != PMC_OK) return failed();
            if (!rf.empty() && pmc_download_radiation_field(ctxs[g], rf.data(), (int64_t)rf.size()) != PMC_OK) return failed();
        }
    };
    {
        std::vector<std::thread> threads;
        for (int g = 1; g < G; ++g) threads.emplace_back(work, g);
        work(0);
        for (auto& t : threads) t.join();
    }
    for (int g = 0; g < G; ++g)
        if (!errors[g].empty())
        {
            fprintf(stderr, "Fatal error (device %d): %s\n", devices[g], errors[g].c_str());
            return 1;
        }
    auto t3 = clock::now();
    pmc_counter_values c{};
    for (int g = 0; g < G; ++g)
    {
        c.cell_visits += counts[g].cell_visits;
        c.detector_updates += counts[g].detector_updates;
        c.stat_overflows += counts[g].stat_overflows;
    }
    printf("Finished primary emission in %.3f s (%.3g packets/s; %.1f cell visits and %.1f detector updates per packet).\n",
           seconds(t2, t3), n / seconds(t2, t3), (double)c.cell_visits / n, (double)c.detector_updates / n);
    bool lost = false;
    for (int g = 0; g < G; ++g)
        if (!statisticsLost[g].empty())
        {
            // FluxRecorder::recordContributions keeps every contribution of a history (FluxRecorder.cpp:962-1014); the engine keeps them
            // in blocks from a pool, and statistics computed from truncated lists would be wrong: the statistics files are not written
            fprintf(stderr, "Warning (device %d): %s\n", devices[g], statisticsLost[g].c_str());
            lost = true;
        }
    if (lost) fprintf(stderr, "Warning: the statistics files (_stats*.fits, _sedstats.dat) are NOT written; the flux files are complete.\n");
    if ((lost ? skh_write_fluxes_only(sim, frames.data(), outdir.c_str()) : skh_write(sim, frames.data(), outdir.c_str())) != 0)
    {
        fprintf(stderr, "Fatal error: %s\n", skh_last_error());
        return 1;
    }
    // the radiation field, if the ski file stores it: summed over the devices (MediumSystem.cpp:1304-1313) and written by
    // the configured RadiationFieldProbe
    if (!rf.empty() && skh_write_radiation_field(sim, rf.data(), outdir.c_str()) != 0)
    {
        fprintf(stderr, "Fatal error: %s\n", skh_last_error());
        return 1;
    }
    printf("Finished final output in %.1f s.\n", seconds(t3, clock::now()));
    for (int g = 0; g < G; ++g)
    {
        pmc_destroy(ctxs[g]);
        pmc_comm_destroy(comms[g]);
    }
    skh_free(sim);
    return 0;
}
