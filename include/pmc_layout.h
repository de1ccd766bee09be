/* pmc_layout.h -- the frame-buffer layout rule of pmc.h as a header-only function, shared by the device library
 * (which exports it as pmc_frame_layout_of), the host model layer and the test oracle so that all three index the
 * detector arrays identically.  Array shapes follow FluxRecorder::finalizeConfiguration
 * (SKIRT/core/FluxRecorder.cpp:185-300) restricted to primary emission without polarisation. */
#ifndef PMC_LAYOUT_H
#define PMC_LAYOUT_H

#include "pmc.h"

static inline int64_t pmc_layout_compute(const pmc_scene* scene, int32_t instrument, pmc_frame_layout* out)
{
    int64_t offset = 0;
    for (int32_t i = 0; i < scene->num_instruments; ++i)
    {
        const pmc_instrument* ins = &scene->instruments[i];
        pmc_frame_layout L;
        L.num_components = ins->record_components ? 3 + ins->num_scattering_levels : 1;
        L.npix = (int64_t)ins->nxp * (int64_t)ins->nyp;
        L.num_lambda = ins->num_lambda;
        int64_t lenSED = ins->include_flux_density ? L.num_lambda : 0;
        int64_t lenIFU = ins->include_surface_brightness ? L.npix * L.num_lambda : 0;
        L.sed_offset = lenSED ? offset : -1;
        offset += L.num_components * lenSED;
        L.ifu_offset = lenIFU ? offset : -1;
        offset += L.num_components * lenIFU;
        L.wsed_offset = (ins->record_statistics && lenSED) ? offset : -1;
        if (ins->record_statistics) offset += 5 * lenSED;
        L.wifu_offset = (ins->record_statistics && lenIFU) ? offset : -1;
        if (ins->record_statistics) offset += 5 * lenIFU;
        L.end_offset = offset;
        if (i == instrument && out) *out = L;
    }
    return offset;
}

#endif
