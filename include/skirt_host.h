/* skirt_host.h -- C entry points of the host model layer (libskirthost.so).
 *
 * The host layer reads an unchanged SKIRT .ski file, performs the reference's setup for the classes the
 * primary-emission path supports (grid construction, density sampling, dust tables, source tables, instruments;
 * SKIRT/core/MonteCarloSimulation.cpp:20-37 setupSimulation) and flattens the result into the pmc_scene of
 * pmc.h.  After the photon loop it calibrates the detector arrays and writes the reference's output files
 * (FluxRecorder::calibrateAndWrite, SKIRT/core/FluxRecorder.cpp:484-846).  It contains NO photon loop.
 */
#ifndef SKIRT_HOST_H
#define SKIRT_HOST_H

#include "pmc.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct skh_simulation skh_simulation;

const char* skh_last_error(void);
/* XmlHierarchyCreator::readFile: parse the ski file; NULL on error */
skh_simulation* skh_load(const char* ski_path);
void skh_free(skh_simulation* sim);
/* overrides applied before skh_setup */
int skh_set_num_packets(skh_simulation* sim, uint64_t n);
int skh_set_tree_topology_file(skh_simulation* sim, const char* treetop_path);
/* before skh_setup: evaluate the densities of an imported particle medium (ParticleMedium) on the GPU during setup.  The
   four pointers are libpmc.so's pmc_sampler_create, pmc_sampler_density, pmc_sampler_destroy and pmc_last_error (passed
   at run time so that this library does not link against the HIP engine); results are bit-identical to the host's */
int skh_set_particle_sampler(skh_simulation* sim, void* create, void* density, void* destroy, void* last_error, int32_t device);
/* Simulation::setupSimulation */
int skh_setup(skh_simulation* sim);
/* valid after skh_setup, owned by the simulation */
const pmc_scene* skh_scene(const skh_simulation* sim);
uint64_t skh_num_packets(const skh_simulation* sim);
int32_t  skh_seed(const skh_simulation* sim);
uint64_t skh_setup_draws(const skh_simulation* sim);
/* luminosity that one launched packet carries at oligochromatic wavelength `index` (SourceSystem.cpp:96,105-106 times the
   weight of NormalizedSource.cpp:91-105); negative if the simulation is not oligochromatic or the index is out of range */
double   skh_packet_luminosity(const skh_simulation* sim, int32_t index);
int64_t  skh_frame_size(const skh_simulation* sim);
int skh_frame_layout(const skh_simulation* sim, int32_t instrument, pmc_frame_layout* out);
/* calibrates `frames` in place and writes <prefix>_<instrument>_*.fits / _sed.dat / _sedstats.dat into outdir */
int skh_write(const skh_simulation* sim, double* frames, const char* outdir);
/* the same without the statistics files (_stats0..4.fits, _sedstats.dat): for a segment whose statistics arrays are incomplete
   (pmc_run_primary returned PMC_ERR_OVERFLOW: the flux arrays are complete, the sums of w^k are not) */
int skh_write_fluxes_only(const skh_simulation* sim, double* frames, const char* outdir);
/* radiation field (RadiationFieldOptions::storeRadiationField): doubles of the table rf[m * nbins + ell] that
   pmc_download_radiation_field fills (0: not stored), and the RadiationFieldProbe / PerCellForm files
   <prefix>_<probe>_J.dat written from it (RadiationFieldProbe.cpp:27-78, PerCellForm.cpp:14-32) */
int64_t  skh_radiation_field_size(const skh_simulation* sim);
int skh_write_radiation_field(const skh_simulation* sim, const double* rf, const char* outdir);
int skh_summary(const skh_simulation* sim, char* buffer, int32_t capacity);

/* A set-up scene as ONE file: everything pmc_create reads plus the numbers a driver of the photon loop needs.  In a job of
   one process per GPU, one process sets the simulation up (all host cores) and saves it, the others load it instead of
   repeating the setup -- the reference repeats Simulation::setupSimulation in every MPI process.  A loaded scene serves
   pmc_create and the frame layout; the output files are written by the process that holds the simulation. */
int skh_scene_save(const skh_simulation* sim, const char* path);
typedef struct skh_scene_file skh_scene_file;
skh_scene_file* skh_scene_load(const char* path);
void skh_scene_file_free(skh_scene_file* file);
const pmc_scene* skh_scene_file_scene(const skh_scene_file* file);
enum { SKH_SCENE_SEED = 0, SKH_SCENE_NUM_PACKETS = 1, SKH_SCENE_FRAME_SIZE = 2, SKH_SCENE_RADIATION_FIELD_SIZE = 3, SKH_SCENE_SETUP_DRAWS = 4 };
int64_t skh_scene_file_number(const skh_scene_file* file, int32_t what);
int skh_scene_file_layout(const skh_scene_file* file, int32_t instrument, pmc_frame_layout* out);

#ifdef __cplusplus
}
#endif
#endif
