// refshim: bmengine/functions/list.h -- ModuleList<T> (list.h:8-37): an owning vector of submodules registered under their index,
// written here against core::Layer of the HIP shim.  Test infrastructure for the compile-the-reference check.
#pragma once
#include <memory>
#include <string>
#include <vector>
#include "bm_functions.h"
#include "bm_layer.h"
namespace bmengine {
namespace functions {
template <typename T>
class ModuleList : public core::Layer {
    std::vector<std::unique_ptr<T>> modules;

public:
    ModuleList() : core::Layer() {}
    ~ModuleList() {
        for (auto it = modules.rbegin(); it != modules.rend(); ++it) it->reset();
    }
    const char* layer_type() const override { return "ModuleList"; }
    template <typename... Params>
    void append(Params&&... params) {
        modules.emplace_back(std::make_unique<T>(std::forward<Params>(params)...));
        add_submodule(std::to_string(modules.size() - 1), *modules.back());
    }
    T& operator[](size_t i) {
        BM_ASSERT(i < modules.size(), "Index out of range");
        return *modules[i];
    }
    size_t size() const { return modules.size(); }
};
}  // namespace functions
}  // namespace bmengine
