// pmc_tuning.hip -- the switch table of include/pmc_tuning.h
#include "pmc_context.h"

// ---- tuning switches (include/pmc_tuning.h): a process-wide table set through pmc_tuning_set; the library reads three settings
// from the environment (PMC_NUM_SLOTS, PMC_NUM_GROUPS, PMC_STAT_POOL_BLOCKS) and nothing else
namespace
{
    std::mutex g_tuneMutex;
    // name -> value; the values live in a pool that is never shrunk, so that a pointer handed out by pmcTune stays valid when another
    // host thread sets or clears switches meanwhile (one host thread per device drives pmc_create / pmc_run_primary in the CLI and in
    // the multi-device tests).  A switch is SAMPLED where it is used -- the table layouts at pmc_create, the kernel selection at
    // pmc_run_primary: changing switches while a context is being created or is running gives that context either value.
    std::map<std::string, const std::string*>& tuneTable()
    {
        static std::map<std::string, const std::string*> table;
        return table;
    }
    const std::string* internTuneValue(const char* value)
    {
        static std::deque<std::string> pool;
        for (const auto& v : pool)
            if (v == value) return &v;
        pool.emplace_back(value);
        return &pool.back();
    }
}
// the value of a tuning switch, or null (the pointer stays valid for the life of the process)
extern "C" const char* pmcTune(const char* name)
{
    std::lock_guard<std::mutex> lock(g_tuneMutex);
    auto& table = tuneTable();
    auto at = table.find(name);
    return at == table.end() ? nullptr : at->second->c_str();
}
extern "C" int pmc_tuning_set(const char* name, const char* value)
{
    if (!name) return PMC_ERR_INVALID;
    std::lock_guard<std::mutex> lock(g_tuneMutex);
    if (value)
        tuneTable()[name] = internTuneValue(value);
    else
        tuneTable().erase(name);
    return PMC_OK;
}
extern "C" void pmc_tuning_clear(void)
{
    std::lock_guard<std::mutex> lock(g_tuneMutex);
    tuneTable().clear();
}
