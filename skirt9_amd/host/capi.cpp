// capi.cpp -- C entry points of the host model layer (libskirthost.so), for ctypes and for the command line driver.
// See include/skirt_host.h for the declarations.

#include "../../include/skirt_host.h"
#include "simulation.hpp"
#include <cstring>
#include <fstream>

namespace
{
    thread_local std::string t_error;
    int fail(const std::exception& e)
    {
        t_error = e.what();
        return -1;
    }
}

// error text of the calling thread, for the other translation units of the library (scenefile.cpp)
const char* skh_set_error_text(const std::string& text)
{
    t_error = text;
    return t_error.c_str();
}

struct skh_simulation
{
    std::unique_ptr<skh::Simulation> sim;
};

extern "C" {

const char* skh_last_error(void)
{
    return t_error.c_str();
}

skh_simulation* skh_load(const char* ski_path)
{
    try
    {
        auto h = new skh_simulation();
        h->sim = skh::Simulation::fromFile(ski_path);
        return h;
    }
    catch (const std::exception& e)
    {
        fail(e);
        return nullptr;
    }
}

void skh_free(skh_simulation* h)
{
    delete h;
}

int skh_set_num_packets(skh_simulation* h, uint64_t n)
{
    h->sim->setNumPackets(n);
    return 0;
}

// Hands the entry points of the device sampler (include/pmc.h: pmc_sampler_create / _density / _destroy, pmc_last_error)
// to the simulation: the densities of an imported particle medium are then evaluated on the GPU during skh_setup.
int skh_set_particle_sampler(skh_simulation* h, void* create, void* density, void* destroy, void* last_error, int32_t device)
{
    try
    {
        if (!h || !h->sim || !create || !density || !destroy) throw std::runtime_error("invalid argument");
        skh::ParticleSamplerApi api;
        api.create = reinterpret_cast<int (*)(const pmc_particles*, int32_t, pmc_sampler**)>(create);
        api.density = reinterpret_cast<int (*)(pmc_sampler*, const double*, int64_t, double*)>(density);
        api.destroy = reinterpret_cast<void (*)(pmc_sampler*)>(destroy);
        api.lastError = reinterpret_cast<const char* (*)()>(last_error);
        api.device = device;
        h->sim->setParticleSampler(api);
        return 0;
    }
    catch (const std::exception& e)
    {
        return fail(e);
    }
}

int skh_set_tree_topology_file(skh_simulation* h, const char* path)
{
    try
    {
        // TreeSpatialGridTopologyProbe format (TreeSpatialGrid.cpp:225-251): comment lines, the number of children
        // of the root, then one "1"/"0" per node in depth-first order
        std::ifstream in(path);
        if (!in) throw std::runtime_error(std::string("cannot open tree topology file ") + path);
        std::vector<char> topology;
        std::string line;
        bool first = true;
        while (std::getline(in, line))
        {
            if (line.empty() || line[0] == '#') continue;
            if (first)
            {
                first = false;
                if (std::stoi(line) != 8 && std::stoi(line) != 0) throw std::runtime_error("topology is not an octree");
                continue;
            }
            topology.push_back(line[0] == '1');
        }
        h->sim->setTreeTopology(std::move(topology));
        return 0;
    }
    catch (const std::exception& e)
    {
        return fail(e);
    }
}

int skh_setup(skh_simulation* h)
{
    try
    {
        h->sim->setup();
        return 0;
    }
    catch (const std::exception& e)
    {
        return fail(e);
    }
}

const pmc_scene* skh_scene(const skh_simulation* h)
{
    return &h->sim->scene();
}

uint64_t skh_num_packets(const skh_simulation* h)
{
    return h->sim->numPackets();
}

int32_t skh_seed(const skh_simulation* h)
{
    return h->sim->seed();
}

double skh_packet_luminosity(const skh_simulation* h, int32_t index)
{
    const pmc_source& src = h->sim->scene().source;
    if (src.lambda_mode != PMC_LAMBDA_OLIGO || index < 0 || index >= src.num_oligo) return -1.;
    return src.packet_luminosity * src.oligo_weight[index];
}

uint64_t skh_setup_draws(const skh_simulation* h)
{
    return h->sim->setupDraws();
}

int64_t skh_frame_size(const skh_simulation* h)
{
    return h->sim->frameSize();
}

int skh_frame_layout(const skh_simulation* h, int32_t instrument, pmc_frame_layout* out)
{
    if (instrument < 0 || instrument >= h->sim->numInstruments()) return -1;
    *out = h->sim->layout(instrument);
    return 0;
}

int skh_write(const skh_simulation* h, double* frames, const char* outdir)
{
    try
    {
        h->sim->write(frames, outdir);
        return 0;
    }
    catch (const std::exception& e)
    {
        return fail(e);
    }
}

// number of doubles of the radiation field table (0: the simulation does not store it)
int64_t skh_radiation_field_size(const skh_simulation* h)
{
    return h && h->sim ? h->sim->radiationFieldSize() : 0;
}

// writes the RadiationFieldProbe files from the table rf[m * nbins + ell]
int skh_write_fluxes_only(const skh_simulation* h, double* frames, const char* outdir)
{
    try
    {
        h->sim->write(frames, outdir, false);
        return 0;
    }
    catch (const std::exception& e)
    {
        return fail(e);
    }
}

int skh_write_radiation_field(const skh_simulation* h, const double* rf, const char* outdir)
{
    try
    {
        if (!h || !h->sim || !rf || !outdir) throw std::runtime_error("invalid argument");
        h->sim->writeRadiationField(rf, outdir);
        return 0;
    }
    catch (const std::exception& e)
    {
        return fail(e);
    }
}

int skh_summary(const skh_simulation* h, char* buffer, int32_t capacity)
{
    std::string s = h->sim->summary();
    if (capacity > 0)
    {
        std::strncpy(buffer, s.c_str(), capacity - 1);
        buffer[capacity - 1] = 0;
    }
    return static_cast<int>(s.size());
}
}
