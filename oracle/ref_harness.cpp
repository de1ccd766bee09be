// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into, imported or called by the product).
//
// A small driver of our own that links the UNMODIFIED reference objects built by oracle/Makefile.ref
// (oracle/_ref/libskirtref.a) and exposes what the parity tests need from the real SKIRT 9 code:
//
//   skirt_ref run  <file.ski> [-t N] [-o outdir] [-e]
//        construct the simulation from the ski file and setupAndRun() it; the counterpart of
//        SKIRT/main/SkirtCommandLineHandler.cpp:295-372 (doSimulation) without the command line front end.
//        Output files are the reference's own (<prefix>_<instr>_total.fits, _sed.dat, _stats*.fits, ...).
//        Prints "PRIMARY_SECONDS <x>" = wall time of runSimulation() (emission + write) and the log line
//        of the reference ("Finished primary emission in ...") goes to <prefix>_log.txt.
//
//   skirt_ref rays <file.ski> <rays.txt> <out.txt>
//        setup in emulation mode (no packets), then for every line "rx ry rz kx ky kz" (k is normalised)
//        of rays.txt walk the reference's PathSegmentGenerator and write "ray i n" followed by n lines
//        "m ds" with ds as a C99 hex float.
//
//   skirt_ref cells <file.ski> <out.txt>
//        setup in emulation mode, then dump per cell: m, box (6 hex floats), number density of medium 0
//        (hex float); plus a header with sectionExt/sectionSca/asymmpar at the wavelengths given with -w.
//
// The harness only calls public member functions of the reference classes.

#include "Configuration.hpp"
#include "FatalError.hpp"
#include "FileLog.hpp"
#include "FilePaths.hpp"
#include "MaterialMix.hpp"
#include "MediumSystem.hpp"
#include "MonteCarloSimulation.hpp"
#include "ParallelFactory.hpp"
#include "PathSegmentGenerator.hpp"
#include "ProcessManager.hpp"
#include "SimulationItemRegistry.hpp"
#include "SpatialGrid.hpp"
#include "StringUtils.hpp"
#include "System.hpp"
#include "XmlHierarchyCreator.hpp"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

namespace
{
    struct Args
    {
        std::string mode, ski, a1, a2, outdir = ".";
        int threads = 1;
        bool emulate = false;
        std::vector<double> wavelengths;
    };

    std::unique_ptr<Item> load(const Args& args, MonteCarloSimulation*& sim, bool emulate)
    {
        auto schema = SimulationItemRegistry::getSchemaDef();
        auto top = XmlHierarchyCreator::readFile(schema, args.ski);
        sim = dynamic_cast<MonteCarloSimulation*>(top.get());
        if (!sim) throw std::runtime_error("top-level item is not a MonteCarloSimulation");
        sim->filePaths()->setOutputPrefix(StringUtils::filenameBase(args.ski));
        sim->filePaths()->setInputPath(".");
        sim->filePaths()->setOutputPath(args.outdir);
        sim->parallelFactory()->setMaxThreadCount(args.threads);
        FileLog* log = new FileLog();
        sim->log()->setLinkedLog(log);
        sim->log()->setLowestLevel(Log::Level::Warning);
        if (emulate) sim->config()->setEmulationMode();
        log->setup();
        return top;
    }
}

int main(int argc, char** argv)
{
    ProcessManager pm(&argc, &argv);
    System system(argc, argv);
    SimulationItemRegistry registry("ref-harness", "9");

    Args args;
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i)
    {
        std::string a = argv[i];
        if (a == "-t" && i + 1 < argc)
            args.threads = atoi(argv[++i]);
        else if (a == "-o" && i + 1 < argc)
            args.outdir = argv[++i];
        else if (a == "-e")
            args.emulate = true;
        else if (a == "-w" && i + 1 < argc)
            args.wavelengths.push_back(atof(argv[++i]));
        else
            pos.push_back(a);
    }
    if (pos.size() < 2)
    {
        fprintf(stderr, "usage: skirt_ref run|rays|cells <file.ski> ...\n");
        return 2;
    }
    args.mode = pos[0];
    args.ski = pos[1];
    if (pos.size() > 2) args.a1 = pos[2];
    if (pos.size() > 3) args.a2 = pos[3];

    try
    {
        MonteCarloSimulation* sim = nullptr;
        if (args.mode == "run")
        {
            auto top = load(args, sim, args.emulate);
            auto t0 = std::chrono::steady_clock::now();
            sim->setupAndRun();
            auto t1 = std::chrono::steady_clock::now();
            printf("TOTAL_SECONDS %.6f\n", std::chrono::duration<double>(t1 - t0).count());
            return 0;
        }
        if (args.mode == "rays")
        {
            auto top = load(args, sim, true);
            sim->setupAndRun();
            auto gen = sim->mediumSystem()->grid()->createPathSegmentGenerator();
            std::ifstream in(args.a1);
            FILE* out = fopen(args.a2.c_str(), "w");
            if (!in || !out) throw std::runtime_error("cannot open ray files");
            std::string line;
            int index = 0;
            while (std::getline(in, line))
            {
                if (line.empty() || line[0] == '#') continue;
                std::istringstream ss(line);
                std::string t[6];
                for (auto& s : t) ss >> s;
                double v[6];
                for (int i = 0; i < 6; ++i) v[i] = strtod(t[i].c_str(), nullptr);  // accepts hex floats
                Position r(v[0], v[1], v[2]);
                Direction k(v[3], v[4], v[5], true);
                gen->start(r, k);
                std::vector<std::pair<int, double>> segs;
                while (gen->next()) segs.emplace_back(gen->m(), gen->ds());
                fprintf(out, "ray %d %zu %a %a %a\n", index++, segs.size(), k.x(), k.y(), k.z());
                for (auto& s : segs) fprintf(out, "%d %a\n", s.first, s.second);
            }
            fclose(out);
            return 0;
        }
        if (args.mode == "cells")
        {
            auto top = load(args, sim, true);
            sim->setupAndRun();
            auto ms = sim->mediumSystem();
            auto grid = ms->grid();
            FILE* out = fopen(args.a1.c_str(), "w");
            if (!out) throw std::runtime_error("cannot open output file");
            int n = ms->numCells();
            fprintf(out, "cells %d\n", n);
            for (double w : args.wavelengths)
            {
                auto mix = ms->mix(0, 0);
                fprintf(out, "mix %a %a %a %a\n", w, mix->sectionExt(w), mix->sectionSca(w), mix->asymmpar(w));
            }
            for (int m = 0; m < n; ++m)
            {
                Position c = grid->centralPositionInCell(m);
                fprintf(out, "%d %a %a %a %a %a\n", m, c.x(), c.y(), c.z(), grid->volume(m), ms->numberDensity(m, 0));
            }
            fclose(out);
            return 0;
        }
        fprintf(stderr, "unknown mode %s\n", args.mode.c_str());
        return 2;
    }
    catch (FatalError& error)
    {
        for (const std::string& line : error.message()) fprintf(stderr, "FatalError: %s\n", line.c_str());
        return 1;
    }
    catch (const std::exception& e)
    {
        fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
}
