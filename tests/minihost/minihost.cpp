// tests/minihost/minihost.cpp -- TEST INFRASTRUCTURE: the smallest host that can load a SatDump plugin and run its modules.
//
// Full SatDump cannot be built in this repository's environment (no volk / fftw3 / nng / ... SURVEY.md 8c), so the plugin
// of plugin/sdhip_plugin.cpp would never execute. This program is the part of SatDump the plugin touches, and nothing else:
//   * it includes the reference's REAL headers (pipeline/module.h, core/plugin.h, utils/event_bus.h, common/dsp/buffer.h) and
//     compiles the reference's own FileStreamToFileStreamModule and dsp buffer constants where they lie (tests/minihost/Makefile);
//   * it defines what libsatdump_core defines for those headers: satdump::eventBus, satdump::pipeline::modules_registry, the
//     handful of ProcessingModule members of src-core/pipeline/module.cpp:36-84 and the registry lookup :123-135 (that file
//     includes every module of SatDump and cannot be compiled alone);
//   * it loads the plugin the way src-core/core/plugin.cpp:15-58 does (dlopen, `loader`, init()), registers stand-ins for the
//     CPU modules under the stock ids, fires RegisterModulesEvent (module.cpp:121) and SatDumpStartedEvent (init.cpp:163), and
//     runs two modules the way Pipeline::run does: one after the other through a file (pipeline_run.cpp:121-180), or both at
//     once joined by a FIFO (pipeline_run.cpp:44-104), or -- as the live pipeline does -- the demodulator fed from a
//     dsp::stream<complex_t> (live_pipeline.cpp:45-105).
// Usage: minihost <libsdhip.so> <plugin.so> list | run <job.json>
#include "core/exception.h"
#include "core/plugin.h"
#include "logger.h"
#include "pipeline/module.h"
#include "pipeline/modules/base/filestream_to_filestream.h"
#include "dsp/block.h"
#include "dsp/flowgraph/dsp_flowgraph_register.h" // RegisterNodesEvent, Flowgraph::NodeInternalReg (the flowgraph-registry variant of the plugin)

#include <dlfcn.h>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <cstring>
#include <thread>

#include "init.h" // ref_shim/init.h: the TLE registry's one method (enable_doppler)

std::shared_ptr<slog::Logger> logger = std::make_shared<slog::Logger>();

// ---- what libsatdump_core provides for the headers above
float ui_scale = 1.0f;
namespace ImGui
{
    ImVec2 GetContentRegionAvail() { return ImVec2(0, 0); }
    void ProgressBar(float, const ImVec2 &, const char *) {}
}
namespace satdump
{
    namespace widgets
    { // node_int.h's inline upd_state() drags the options GUI's inline code in, which refers to this widget (common/widgets/double_list.cpp: ImGui): never
      // constructed here (no node gets an options displayer), defined so that the host links
        DoubleList::DoubleList(std::string name) : d_id(name), current_value(nullptr) {}
        DoubleList::~DoubleList() {}
        void DoubleList::set_list(std::vector<double> list, bool, std::string) { available_values = list; }
        bool DoubleList::set_value(double, double) { return false; }
    } // namespace widgets
    namespace ndsp
    {
        namespace flowgraph
        {
            // src-core/dsp/flowgraph/node_int.cpp without its GUI side (the OptDisplayerWarper it creates and render()'s ImGui calls): what the registry entries
            // of dsp_flowgraph_register.h:23-28 -- and the plugin's RegisterNodesEvent handler -- construct around a block
            NodeInternal::NodeInternal(const Flowgraph *f, std::shared_ptr<ndsp::Block> b) : f((Flowgraph *)f), blk(b) {}
            bool NodeInternal::render() { return false; }
            nlohmann::json NodeInternal::getP() { return blk->get_cfg(); }
            void NodeInternal::setP(nlohmann::json p) { blk->set_cfg(p); }
        } // namespace flowgraph
    } // namespace ndsp
    uint64_t getFilesize(std::string filepath) { return std::filesystem::exists(filepath) ? (uint64_t)std::filesystem::file_size(filepath) : 0; }
    std::map<std::string, std::shared_ptr<satdump::Plugin>> loaded_plugins;
    std::shared_ptr<EventBus> eventBus = std::make_shared<EventBus>();
    std::shared_ptr<KeplerDBHandler> db_keplers = std::make_shared<KeplerDBHandler>(); // filled from the job's "tles" (libsatdump_core: the TLE database)
    std::shared_ptr<TaskScheduler> taskScheduler;
    namespace pipeline
    {
        // src-core/pipeline/module.cpp:36-84
        ProcessingModule::ProcessingModule(std::string input_file, std::string output_file_hint, nlohmann::json parameters)
            : d_parameters(parameters), d_input_file(input_file), d_output_file_hint(output_file_hint)
        {
            input_active = false;
            d_is_streaming_input = false;
        }
        std::string ProcessingModule::getOutput() { return d_output_file; }
        void ProcessingModule::setInputType(ModuleDataType type)
        {
            input_data_type = type;
            d_is_streaming_input = type != DATA_FILE;
            bool found = false;
            for (auto &t : getInputTypes())
                found |= t == type;
            if (!found)
                throw satdump_exception("Module input type not supported! (" + getIDM() + ") : " + std::to_string(type));
        }
        void ProcessingModule::setOutputType(ModuleDataType type)
        {
            output_data_type = type;
            bool found = false;
            for (auto &t : getOutputTypes())
                found |= t == type;
            if (!found)
                throw satdump_exception("Module output type not supported!");
        }
        ModuleDataType ProcessingModule::getInputType() { return input_data_type; }
        ModuleDataType ProcessingModule::getOutputType() { return output_data_type; }
        void ProcessingModule::init() {}
        void ProcessingModule::stop() {}
        void ProcessingModule::drawUI(bool) {}
        std::vector<ModuleEntry> modules_registry;
        std::shared_ptr<ProcessingModule> getModuleInstance(std::string id, std::string input_file, std::string output_file_hint, nlohmann::json parameters)
        { // module.cpp:123-129: first match
            for (auto &m : modules_registry)
                if (m.id == id)
                    return m.inst(input_file, output_file_hint, parameters);
            throw satdump_exception("Could not find module " + id);
        }
    }
}

using namespace satdump::pipeline;

// Stand-in for a CPU module of the stock registry (psk_demod & co. are registered before the plugins' handlers fire,
// module.cpp:91-121). It only says who it is: a job that ends up here was NOT taken over by the plugin.
template <int WHICH>
class CpuStandIn : public ProcessingModule
{
public:
    CpuStandIn(std::string i, std::string o, nlohmann::json p) : ProcessingModule(i, o, p) {}
    std::vector<ModuleDataType> getInputTypes() { return {DATA_FILE, DATA_STREAM, DATA_DSP_STREAM}; }
    std::vector<ModuleDataType> getOutputTypes() { return {DATA_FILE, DATA_STREAM}; }
    void process() { throw satdump_exception("cpu stand-in invoked"); }
    void drawUI(bool) {}
    static const char *name()
    {
        static const char *n[9] = {"psk_demod", "ccsds_conv_concat_decoder", "metop_ahrpt_decoder", "ccsds_simple_psk_decoder", "dvbs2_demod", "meteor_lrpt_decoder", "fengyun_ahrpt_decoder",
                                   "fengyun_mpt_decoder", "dvbs2_ts_extractor"};
        return n[WHICH];
    }
    static std::string getID() { return name(); }
    std::string getIDM() { return std::string("cpu:") + name(); }
    static nlohmann::json getParams() { return {}; }
    static std::shared_ptr<ProcessingModule> getInstance(std::string i, std::string o, nlohmann::json p) { return std::make_shared<CpuStandIn<WHICH>>(i, o, p); }
};

static int fail(const std::string &m)
{
    std::cerr << "minihost: " << m << std::endl;
    return 2;
}

int main(int argc, char **argv)
{
    if (argc < 4)
        return fail("usage: minihost <libsdhip.so> <plugin.so> list | run <job.json>");
    // the C-ABI library first, globally, so that the plugin's sdhip_* references resolve (in a SatDump tree the plugin links it)
    if (!dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL))
        return fail(std::string("dlopen ") + argv[1] + ": " + dlerror());
    // src-core/core/plugin.cpp:15-35
    void *dyn = dlopen(argv[2], RTLD_LAZY);
    if (!dyn)
        return fail(std::string("dlopen ") + argv[2] + ": " + dlerror());
    void *create = dlsym(dyn, "loader");
    if (!create)
        return fail("plugin has no loader()");
    std::shared_ptr<satdump::Plugin> plugin(reinterpret_cast<satdump::Plugin *(*)()>(create)());
    plugin->init();
    satdump::loaded_plugins[plugin->getID()] = plugin;
    // module.cpp:91-121: core modules, then the plugins'
    REGISTER_MODULE(CpuStandIn<0>);
    REGISTER_MODULE(CpuStandIn<1>);
    REGISTER_MODULE(CpuStandIn<3>);
    satdump::eventBus->fire_event<RegisterModulesEvent>({modules_registry});
    REGISTER_MODULE(CpuStandIn<2>); // metop_ahrpt_decoder comes from ANOTHER plugin, possibly loaded after ours (SURVEY 8b ordering caveat)
    REGISTER_MODULE(CpuStandIn<4>); // dvbs2_demod likewise (plugins/dvb_support)
    REGISTER_MODULE(CpuStandIn<5>); // meteor_lrpt_decoder (plugins/meteor_support)
    REGISTER_MODULE(CpuStandIn<6>); // fengyun_ahrpt_decoder (plugins/fengyun3_support)
    REGISTER_MODULE(CpuStandIn<7>); // fengyun_mpt_decoder (the same plugin)
    REGISTER_MODULE(CpuStandIn<8>); // dvbs2_ts_extractor (plugins/dvb_support)
    satdump::eventBus->fire_event<satdump::SatDumpStartedEvent>({}); // init.cpp:163

    const std::string cmd = argv[3];
    if (cmd == "list")
    {
        std::cout << plugin->getID();
        for (auto &m : modules_registry)
            std::cout << " " << m.id;
        std::cout << std::endl;
        return 0;
    }
    if (cmd == "ndsp" && argc >= 5)
    { // the ndsp block of the plugin between two DSPStream FIFOs, fed and drained like a flowgraph would (src-core/dsp/block.h: set_input /
      // get_output / start / stop; terminator propagation as in block_simple.h:31-37): cf32 file in, cf32 symbol file out
        using namespace satdump::ndsp;
        nlohmann::ordered_json job;
        {
            std::ifstream f(argv[4]);
            f >> job;
        }
        try
        {
            nlohmann::json report;
            std::shared_ptr<Block> blk;
            if (job.value("via_registry", false))
            { // the flowgraph's node registry (dsp_flowgraph_register.cpp: the stock nodes first, then RegisterNodesEvent :438): stand-ins under the stock
              // ids, the plugin's handler adds its nodes (and, under SDHIP_OVERRIDE=1 with a device, re-points the stock ones); the node is then made
              // the way Flowgraph::addNode does it -- registry entry's func(flowgraph) -- and its block run below
                using namespace satdump::ndsp::flowgraph;
                std::map<std::string, Flowgraph::NodeInternalReg> reg;
                for (const char *id : {"psk_demod_cc", "rrc_fir_cc", "agc_cc", "clock_recovery_mm_cc", "costas_cc", "clock_recovery_gardner_cc", "agc_fast_cc", "costas_fast_cc",
                                       "fast_clock_recovery_mm_cc"})
                    reg.insert({id, {std::string("stock/") + id, [](const Flowgraph *) { return std::shared_ptr<NodeInternal>(); }}});
                satdump::eventBus->fire_event<RegisterNodesEvent>({reg});
                nlohmann::json ids = nlohmann::json::object();
                for (auto &kv : reg)
                    ids[kv.first] = kv.second.menuname;
                report["registry"] = ids;
                const std::string id = job["block"].get<std::string>();
                if (!reg.count(id))
                    return fail("node id not in the registry: " + id);
                std::shared_ptr<NodeInternal> node = reg.at(id).func(nullptr);
                if (!node)
                {
                    report["node"] = "stock stand-in";
                    std::cout << report.dump() << std::endl;
                    return 0;
                }
                blk = node->blk;
                report["node_cfg"] = node->getP();
            }
            else
            {
                auto make = reinterpret_cast<Block *(*)(const char *)>(dlsym(dyn, "sdhip_plugin_make_ndsp_block"));
                if (!make)
                    return fail("plugin has no sdhip_plugin_make_ndsp_block()");
                blk.reset(make(job["block"].get<std::string>().c_str()));
            }
            if (!blk)
                return fail("unknown ndsp block");
            report["block"] = blk->d_id;
            nlohmann::json res = nlohmann::json::object();
            for (auto &kv : job["cfg"].items())
                res[kv.key()] = (int)blk->set_cfg(kv.key(), nlohmann::json(kv.value()));
            report["set_cfg"] = res;
            std::vector<float> in;
            {
                std::ifstream f(job["input"].get<std::string>(), std::ios::binary | std::ios::ate);
                in.resize((size_t)f.tellg() / sizeof(float));
                f.seekg(0);
                f.read((char *)in.data(), (std::streamsize)in.size() * sizeof(float));
            }
            const size_t n = in.size() / 2, buf = job.value("buffer", 8192);
            BlockIO src{"in", DSP_SAMPLE_TYPE_CF32};
            src.fifo = std::make_shared<DSPStream>(4);
            blk->set_input(src, 0);
            BlockIO dst = blk->get_output(0, 4);
            blk->start();
            std::thread feeder(
                [&]()
                {
                    for (size_t o = 0; o < n; o += buf)
                    {
                        const size_t m = std::min(buf, n - o);
                        DSPBuffer b = src.fifo->newBufferSamples((uint32_t)buf, sizeof(complex_t));
                        memcpy(b.getSamples<complex_t>(), in.data() + 2 * o, m * sizeof(complex_t));
                        b.size = (uint32_t)m;
                        src.fifo->wait_enqueue(b);
                    }
                    src.fifo->wait_enqueue(src.fifo->newBufferTerminator());
                });
            std::ofstream out(job["output"].get<std::string>(), std::ios::binary);
            size_t got = 0;
            for (;;)
            {
                DSPBuffer b = dst.fifo->wait_dequeue();
                if (b.isTerminator())
                {
                    dst.fifo->free(b);
                    break;
                }
                out.write((const char *)b.getSamples<complex_t>(), (std::streamsize)b.size * sizeof(complex_t));
                got += b.size;
                dst.fifo->free(b);
            }
            feeder.join();
            report["cfg_list"] = blk->get_cfg_list();
            if (report["cfg_list"].contains("pll_freq"))
                report["pll_freq"] = blk->get_cfg("pll_freq");
            blk->stop();
            report["symbols"] = got;
            std::cout << report.dump() << std::endl;
        }
        catch (const std::exception &e)
        {
            return fail(std::string("exception: ") + e.what());
        }
        return 0;
    }
    if (cmd != "run" || argc < 5)
        return fail("unknown command");
    nlohmann::json job;
    {
        std::ifstream f(argv[4]);
        f >> job;
    }
    if (job.contains("tles"))
        for (auto &e : job["tles"])
            satdump::db_keplers->tles.push_back(e.get<satdump::TLE>());
    try
    {
        const std::string mode = job["mode"], input = job["input"], hint = job["output_hint"];
        auto m1 = getModuleInstance(job["demod"]["module"], input, hint, job["demod"]["parameters"]);
        nlohmann::json report;
        report["demod_class"] = m1->getIDM();
        std::shared_ptr<ProcessingModule> m2;
        if (job.contains("decoder"))
        {
            m2 = getModuleInstance(job["decoder"]["module"], input, hint, job["decoder"]["parameters"]);
            report["decoder_class"] = m2->getIDM();
        }
        if (job.value("instantiate_only", false))
        {
            std::cout << report.dump() << std::endl;
            return 0;
        }
        if (mode == "file")
        { // pipeline_run.cpp:121-180: one module after the other, the output file of one is the input of the next
            m1->setInputType(DATA_FILE);
            m1->setOutputType(DATA_FILE);
            m1->init();
            m1->process();
            report["soft"] = m1->getOutput();
            report["demod_stats"] = m1->getModuleStats();
            if (m2)
            {
                m2 = getModuleInstance(job["decoder"]["module"], m1->getOutput(), hint, job["decoder"]["parameters"]);
                m2->setInputType(DATA_FILE);
                m2->setOutputType(DATA_FILE);
                m2->init();
                m2->process();
                report["cadu"] = m2->getOutput();
                report["decoder_stats"] = m2->getModuleStats();
            }
        }
        else if (mode == "fifo" || mode == "dsp_stream")
        {
            if (!m2)
                return fail("fifo / dsp_stream need a decoder");
            std::thread feeder;
            if (mode == "fifo")
                m1->setInputType(DATA_FILE);
            else
            { // live_pipeline.cpp: the source's dsp::stream is the demodulator's input
                m1->setInputType(DATA_DSP_STREAM);
                m1->input_stream = std::make_shared<dsp::stream<complex_t>>();
                m1->input_active = true;
            }
            // pipeline_run.cpp:72-104
            m1->setOutputType(DATA_STREAM);
            m1->output_fifo = std::make_shared<dsp::RingBuffer<uint8_t>>(1000000);
            m2->input_fifo = m1->output_fifo;
            m2->setInputType(DATA_STREAM);
            m2->setOutputType(DATA_FILE);
            m2->input_active = true;
            m1->init();
            m2->init();
            std::thread module1_thread([&m1]() { m1->process(); });
            std::thread module2_thread([&m2]() { m2->process(); });
            if (mode == "dsp_stream")
            {
                const int chunk = job.value("source_buffer", 8192);
                std::ifstream f(input, std::ios::binary);
                while (f)
                {
                    f.read((char *)m1->input_stream->writeBuf, (std::streamsize)chunk * sizeof(complex_t));
                    const int got = (int)(f.gcount() / sizeof(complex_t));
                    if (got <= 0)
                        break;
                    if (!m1->input_stream->swap(got))
                        break;
                }
                // an empty hand-off: swap() only returns once the reader has flushed the previous buffer (buffer.h:49-106), i.e. once
                // the demodulator has taken the last real one -- then end the stream like a stopped source does
                m1->input_stream->swap(0);
                m1->input_active = false;
                m1->input_stream->stopWriter();
                m1->input_stream->stopReader();
                m1->stop();
            }
            if (module1_thread.joinable())
                module1_thread.join();
            while (m2->input_fifo->getReadable() > 0)
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
            m2->input_active = false;
            m2->input_fifo->stopReader();
            m2->input_fifo->stopWriter();
            m2->stop();
            if (module2_thread.joinable())
                module2_thread.join();
            report["cadu"] = m2->getOutput();
            report["demod_stats"] = m1->getModuleStats();
            report["decoder_stats"] = m2->getModuleStats();
        }
        else
            return fail("unknown mode " + mode);
        std::cout << report.dump() << std::endl;
    }
    catch (const std::exception &e)
    {
        return fail(std::string("exception: ") + e.what());
    }
    return 0;
}
