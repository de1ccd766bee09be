// scenefile.cpp -- a set-up scene as ONE file: everything pmc_create reads (include/pmc.h pmc_scene: grid, densities, dust tables,
// sources, instruments) plus the few numbers a driver of the photon loop needs (seed, packets of the segment, frame layout).
//
// In a job of one process per GPU every process needs the same scene.  The reference repeats the whole setup in every MPI
// process (Simulation::setupSimulation runs everywhere; only the photon packets are distributed, SKIRT/mpi/ProcessManager.cpp).
// Here ONE process sets the simulation up with all host cores, saves the scene (to /dev/shm, say), and the others load it:
// no second tree construction, no second density sampling.  The file is the pmc_scene with every pointer replaced by an
// offset into the file; loading reads it into one allocation and turns the offsets back into pointers.

#include "../../include/skirt_host.h"
#include "simulation.hpp"
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

extern const char* skh_set_error_text(const std::string& text);  // capi.cpp

namespace
{
    constexpr uint64_t MAGIC = 0x32454e454353484bull;  // "KHSCENE2"

    // The loader trusts nothing in the file: the header names the sizes of the structures the writer was built with and a checksum
    // of everything behind it; every member's [offset, offset + count * sizeof) is held against the file size before the offset
    // becomes a pointer, and a count is only ever read through a member that passed that test.
    struct Header
    {
        uint64_t magic;
        int32_t abi, seed;
        uint64_t num_packets, setup_draws;
        int64_t frame_size, rf_size;
        int32_t num_instruments, pad;
        uint64_t scene_offset, layout_offset, total_bytes;
        uint32_t sizeof_scene, sizeof_instrument, sizeof_source, sizeof_layout;
        uint64_t checksum;  // FNV-1a (64 bit) of the bytes behind the header
    };

    uint64_t fnv1a(const char* data, size_t bytes)
    {
        // eight interleaved lanes over 8-byte words (a scene file is tens to hundreds of MB), folded at the end
        uint64_t h[8];
        for (int k = 0; k < 8; ++k) h[k] = 0xcbf29ce484222325ull + uint64_t(k);
        size_t i = 0;
        for (; i + 64 <= bytes; i += 64)
            for (int k = 0; k < 8; ++k)
            {
                uint64_t w;
                std::memcpy(&w, data + i + 8 * k, 8);
                h[k] = (h[k] ^ w) * 0x100000001b3ull;
            }
        uint64_t r = 0xcbf29ce484222325ull;
        for (int k = 0; k < 8; ++k) r = (r ^ h[k]) * 0x100000001b3ull;
        for (; i < bytes; ++i) r = (r ^ uint64_t((unsigned char)data[i])) * 0x100000001b3ull;
        return r;
    }

    // Every pointer member of a pmc_scene, in an order in which the element count of each is known from members visited before
    // it: f(pointer member, number of elements).  Used with the caller's arrays (save: append the data) and with offsets (load:
    // rebase), so the counts are always read through pointers that are valid at that moment.
    template<typename F> void visitSource(pmc_source& s, F&& f)
    {
        f(s.sersic_s, s.kind == PMC_SOURCE_SERSIC ? size_t(s.sersic_n) : 0);
        f(s.sersic_M, s.kind == PMC_SOURCE_SERSIC ? size_t(s.sersic_n) : 0);
        f(s.oligo_lambda, s.lambda_mode == PMC_LAMBDA_OLIGO ? size_t(s.num_oligo) : 0);
        f(s.oligo_weight, s.lambda_mode == PMC_LAMBDA_OLIGO ? size_t(s.num_oligo) : 0);
        f(s.sed_lambda, s.lambda_mode == PMC_LAMBDA_TABULATED ? size_t(s.num_sed) : 0);
        f(s.sed_p, s.lambda_mode == PMC_LAMBDA_TABULATED ? size_t(s.num_sed) : 0);
        f(s.sed_P, s.lambda_mode == PMC_LAMBDA_TABULATED ? size_t(s.num_sed) : 0);
    }

    template<typename F> void visitScene(pmc_scene& s, F&& f)
    {
        pmc_grid& g = s.grid;
        const bool cart = g.kind == PMC_GRID_CARTESIAN, tree = g.kind == PMC_GRID_OCTREE, voro = g.kind == PMC_GRID_VORONOI;
        f(g.xv, cart ? size_t(g.nx) + 1 : 0);
        f(g.yv, cart ? size_t(g.ny) + 1 : 0);
        f(g.zv, cart ? size_t(g.nz) + 1 : 0);
        f(g.node_box, tree ? 6 * size_t(g.num_nodes) : 0);
        f(g.node_level, tree ? size_t(g.num_nodes) : 0);
        f(g.node_first_child, tree ? size_t(g.num_nodes) : 0);
        f(g.node_cell, tree ? size_t(g.num_nodes) : 0);
        f(g.nbr_start, tree ? 6 * size_t(g.num_nodes) + 1 : 0);
        f(g.nbr_list, tree ? size_t(g.nbr_start[6 * size_t(g.num_nodes)]) : 0);
        f(g.site, voro ? 3 * size_t(g.num_cells) : 0);
        f(g.vnbr_start, voro ? size_t(g.num_cells) + 1 : 0);
        f(g.vnbr_list, voro ? size_t(g.vnbr_start[g.num_cells]) : 0);
        const size_t blocks = voro ? size_t(g.vblock_n) * g.vblock_n * g.vblock_n : 0;
        f(g.vblock_start, voro ? blocks + 1 : 0);
        f(g.vblock_list, voro ? size_t(g.vblock_start[blocks]) : 0);
        auto visitMedium = [&](pmc_medium& m) {
            f(m.number_density, size_t(g.num_cells));
            f(m.lambda_border, size_t(m.num_lambda));
            f(m.sigma_ext, size_t(m.num_lambda));
            f(m.sigma_sca, size_t(m.num_lambda));
            f(m.asymmpar, size_t(m.num_lambda));
            f(m.sigma_abs, size_t(m.num_lambda));
        };
        visitMedium(s.medium);
        visitSource(s.source, f);
        f(s.instruments, size_t(s.num_instruments));
        for (int i = 0; i < s.num_instruments; ++i)
        {
            pmc_instrument& I = const_cast<pmc_instrument&>(s.instruments[i]);
            f(I.border, size_t(I.num_border));
            f(I.ellv, size_t(I.num_border) + 1);
        }
        pmc_radiation_field& r = s.radiation_field;
        f(r.border, r.store ? size_t(r.num_border) : 0);
        f(r.ellv, r.store ? size_t(r.num_border) + 1 : 0);
        const size_t ns = s.num_sources > 1 ? size_t(s.num_sources) : 0;
        f(s.sources, ns);
        for (size_t i = 0; i < ns; ++i) visitSource(const_cast<pmc_source&>(s.sources[i]), f);
        f(s.source_first, ns ? ns + 1 : 0);
        const size_t nm = s.num_media > 1 ? size_t(s.num_media) : 0;
        f(s.media, nm);
        for (size_t i = 0; i < nm; ++i) visitMedium(const_cast<pmc_medium&>(s.media[i]));
    }

    // save: the data of a member goes to the end of the blob, and the COPY of the structure that holds the member (it lives in
    // the blob as well) gets the offset in place of the pointer
    struct Saver
    {
        std::vector<char> blob;
        size_t append(const void* data, size_t bytes)
        {
            const size_t at = (blob.size() + 15) & ~size_t(15);
            blob.resize(at + bytes);
            if (bytes) std::memcpy(blob.data() + at, data, bytes);
            return at;
        }
    };
}

struct skh_scene_file
{
    std::vector<char> blob;
    const Header* header{nullptr};
    const pmc_scene* scene{nullptr};
    const pmc_frame_layout* layouts{nullptr};
};

extern "C" {

int skh_scene_save(const skh_simulation* h, const char* path)
{
    try
    {
        if (!h || !path) throw std::runtime_error("skh_scene_save: invalid argument");
        const pmc_scene& live = *skh_scene(h);
        Saver S;
        S.blob.resize(sizeof(Header));
        const size_t sceneAt = S.append(&live, sizeof(pmc_scene));
        // The members are visited on the LIVE scene (valid pointers, from which the counts are read); what is patched is the copy
        // in the blob.  Nested structures (instruments, sources) are copied first and then patched at their place in the blob.
        pmc_scene walk = live;  // (a scratch copy whose pointer members are redirected to blob copies where nested members follow)
        std::vector<std::pair<size_t, size_t>> fix;  // (position of a pointer member in the blob, offset of its data)
        // position of a member inside the blob: members of `walk` map to the scene copy, members of nested arrays to their copies
        struct Region
        {
            const char* liveBase;
            size_t bytes, blobAt;
        };
        std::vector<Region> regions{{reinterpret_cast<const char*>(&walk), sizeof(pmc_scene), sceneAt}};
        auto place = [&](const void* member) -> size_t {
            const char* p = static_cast<const char*>(member);
            for (const Region& r : regions)
                if (p >= r.liveBase && p < r.liveBase + r.bytes) return r.blobAt + size_t(p - r.liveBase);
            throw std::runtime_error("skh_scene_save: member outside the known structures");
        };
        visitScene(walk, [&](auto& member, size_t count) {
            typedef typename std::remove_const<typename std::remove_pointer<typename std::remove_reference<decltype(member)>::type>::type>::type T;
            const size_t bytes = count * sizeof(T);
            const size_t at = (member && bytes) ? S.append(member, bytes) : 0;
            fix.emplace_back(place(&member), (member && bytes) ? at : 0);
            // nested structures: their own pointer members are visited next, through the live array
            if ((std::is_same<T, pmc_instrument>::value || std::is_same<T, pmc_source>::value || std::is_same<T, pmc_medium>::value) && member && bytes)
                regions.push_back({reinterpret_cast<const char*>(member), bytes, at});
        });
        for (const auto& f : fix)
        {
            uint64_t off = f.second;
            std::memcpy(S.blob.data() + f.first, &off, sizeof(off));
        }
        std::vector<pmc_frame_layout> layouts(size_t(live.num_instruments));
        for (int i = 0; i < live.num_instruments; ++i) skh_frame_layout(h, i, &layouts[size_t(i)]);
        const size_t layoutAt = S.append(layouts.data(), layouts.size() * sizeof(pmc_frame_layout));
        Header head{};
        head.magic = MAGIC;
        head.abi = PMC_ABI_VERSION;
        head.seed = skh_seed(h);
        head.num_packets = skh_num_packets(h);
        head.setup_draws = skh_setup_draws(h);
        head.frame_size = skh_frame_size(h);
        head.rf_size = skh_radiation_field_size(h);
        head.num_instruments = live.num_instruments;
        head.scene_offset = sceneAt;
        head.layout_offset = layoutAt;
        head.total_bytes = S.blob.size();
        head.sizeof_scene = sizeof(pmc_scene), head.sizeof_instrument = sizeof(pmc_instrument), head.sizeof_source = sizeof(pmc_source);
        head.sizeof_layout = sizeof(pmc_frame_layout);
        head.checksum = fnv1a(S.blob.data() + sizeof(Header), S.blob.size() - sizeof(Header));
        std::memcpy(S.blob.data(), &head, sizeof(head));
        // (written under a temporary name and renamed: a reader never sees a partial file)
        const std::string tmp = std::string(path) + ".part";
        FILE* out = std::fopen(tmp.c_str(), "wb");
        if (!out) throw std::runtime_error("cannot write " + tmp);
        const bool ok = std::fwrite(S.blob.data(), 1, S.blob.size(), out) == S.blob.size();
        if (std::fclose(out) != 0 || !ok) throw std::runtime_error("short write to " + tmp);
        if (std::rename(tmp.c_str(), path) != 0) throw std::runtime_error(std::string("cannot rename to ") + path);
        return 0;
    }
    catch (const std::exception& e)
    {
        skh_set_error_text(e.what());
        return -1;
    }
}

skh_scene_file* skh_scene_load(const char* path)
{
    try
    {
        if (!path) throw std::runtime_error("skh_scene_load: invalid argument");
        FILE* in = std::fopen(path, "rb");
        if (!in) throw std::runtime_error(std::string("cannot read ") + path);
        std::fseek(in, 0, SEEK_END);
        const long size = std::ftell(in);
        std::fseek(in, 0, SEEK_SET);
        auto file = std::unique_ptr<skh_scene_file>(new skh_scene_file());
        file->blob.resize(size > 0 ? size_t(size) : 0);
        const bool ok = size > 0 && std::fread(file->blob.data(), 1, size_t(size), in) == size_t(size);
        std::fclose(in);
        if (!ok || size_t(size) < sizeof(Header)) throw std::runtime_error(std::string("short read from ") + path);
        char* base = file->blob.data();
        const Header* head = reinterpret_cast<const Header*>(base);
        if (head->magic != MAGIC || head->total_bytes != uint64_t(size)) throw std::runtime_error(std::string(path) + " is not a scene file");
        if (head->abi != PMC_ABI_VERSION || head->sizeof_scene != sizeof(pmc_scene) || head->sizeof_instrument != sizeof(pmc_instrument)
            || head->sizeof_source != sizeof(pmc_source) || head->sizeof_layout != sizeof(pmc_frame_layout))
            throw std::runtime_error(std::string(path) + " was written for another version of pmc_scene");
        if (head->checksum != fnv1a(base + sizeof(Header), size_t(size) - sizeof(Header)))
            throw std::runtime_error(std::string(path) + ": checksum mismatch (truncated or corrupt scene file)");
        const uint64_t total = uint64_t(size);
        auto inside = [&](uint64_t off, uint64_t count, uint64_t each) { return off >= sizeof(Header) && off % 8 == 0 && count <= total / each && off <= total - count * each; };
        if (head->num_instruments < 0 || head->num_instruments > 4096 || !inside(head->scene_offset, 1, sizeof(pmc_scene))
            || !inside(head->layout_offset, uint64_t(head->num_instruments), sizeof(pmc_frame_layout)))
            throw std::runtime_error(std::string(path) + ": header offsets outside the file");
        pmc_scene& scene = *reinterpret_cast<pmc_scene*>(base + head->scene_offset);
        if (scene.num_instruments != head->num_instruments) throw std::runtime_error(std::string(path) + ": inconsistent instrument count");
        // offsets back into pointers, in the visiting order (the count of a member is read through members rebased -- and bounds-checked
        // -- before it; a negative count arrives here as a huge one and fails the test)
        visitScene(scene, [&](auto& member, size_t count) {
            typedef typename std::remove_reference<decltype(member)>::type P;
            typedef typename std::remove_const<typename std::remove_pointer<P>::type>::type T;
            uint64_t off = 0;
            std::memcpy(&off, &member, sizeof(off));
            if (count == 0)
            {
                member = nullptr;
                return;
            }
            if (!inside(off, count, sizeof(T))) throw std::runtime_error(std::string(path) + ": a table of the scene lies outside the file");
            member = reinterpret_cast<P>(base + off);
        });
        file->header = head;
        file->scene = &scene;
        file->layouts = reinterpret_cast<const pmc_frame_layout*>(base + head->layout_offset);
        return file.release();
    }
    catch (const std::exception& e)
    {
        skh_set_error_text(e.what());
        return nullptr;
    }
}

void skh_scene_file_free(skh_scene_file* file)
{
    delete file;
}

const pmc_scene* skh_scene_file_scene(const skh_scene_file* file)
{
    return file ? file->scene : nullptr;
}

int64_t skh_scene_file_number(const skh_scene_file* file, int32_t what)
{
    if (!file) return -1;
    switch (what)
    {
        case SKH_SCENE_SEED: return file->header->seed;
        case SKH_SCENE_NUM_PACKETS: return (int64_t)file->header->num_packets;
        case SKH_SCENE_FRAME_SIZE: return file->header->frame_size;
        case SKH_SCENE_RADIATION_FIELD_SIZE: return file->header->rf_size;
        case SKH_SCENE_SETUP_DRAWS: return (int64_t)file->header->setup_draws;
        default: return -1;
    }
}

int skh_scene_file_layout(const skh_scene_file* file, int32_t instrument, pmc_frame_layout* out)
{
    if (!file || !out || instrument < 0 || instrument >= file->header->num_instruments) return -1;
    *out = file->layouts[instrument];
    return 0;
}
}
