#include "dist.hpp"

namespace lashhost {

std::string run_dist(const DistOptions &)
{
    return "lash dist is not built yet in the gfx950 port (SURVEY.md section 8(f), row f2); "
           "the sketch files written by `lash sketch` are the reference's format and can be read by upstream `lash dist`";
}

}  // namespace lashhost
