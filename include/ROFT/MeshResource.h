// ROFT::MeshResource -- the text of an object's mesh (src/roft-lib/include/ROFT/MeshResource.h:23-47, src/MeshResource.cpp:19-62):
// from the "internal data base" <set>/<name>.obj -- compiled into the reference's library as resources; this library embeds no
// meshes, the data base is the directory the environment variable ROFT_MESH_DB names (the reference's src/roft-lib/meshes has
// that layout) -- or from ModelParameters::mesh_external_path.
#pragma once

#include <cstdlib>
#include <fstream>
#include <sstream>

#include "Sources.h"

namespace ROFT {

class MeshResource {
public:
    MeshResource(const std::string& name, const std::string& set) { load_internal(name, set); }
    explicit MeshResource(const ModelParameters& model_parameters)
    {
        if (model_parameters.use_internal_db()) load_internal(model_parameters.name(), model_parameters.internal_db_name());
        else if (!read(model_parameters.mesh_external_path()))
            throw std::runtime_error(log_name_ + "::ctor. Cannot open model from external path " + model_parameters.mesh_external_path() + ".");
    }
    virtual ~MeshResource() = default;
    const std::string& as_string() const { return data_; }

private:
    void load_internal(const std::string& name, const std::string& set)
    {
        const char* db = std::getenv("ROFT_MESH_DB");
        if (!db || !read(std::string(db) + "/" + set + "/" + name + ".obj"))
            throw std::runtime_error(log_name_ + "::ctor. Cannot find requested mesh among available resources (no meshes are compiled into this "
                                                 "library: ROFT_MESH_DB names a directory holding <set>/<name>.obj; or model.use_internal_db = false "
                                                 "with model.external_path).");
    }
    bool read(const std::string& path)
    {
        std::ifstream in(path);
        if (!in.is_open()) return false;
        std::stringstream buffer;
        buffer << in.rdbuf();
        data_ = buffer.str();
        return true;
    }
    std::string data_;
    const std::string log_name_ = "MeshResource";
};

}  // namespace ROFT
