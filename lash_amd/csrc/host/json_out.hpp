// json_out.hpp — the two small JSON documents `lash sketch` writes, byte-for-byte as serde_json's pretty printer
// emits them: {o}_files.json = to_writer_pretty(&files) (/root/reference/src/utils.rs:577-580) and
// {o}_parameters.json = to_string_pretty(json!({...})) with sorted keys and string values (main.rs:254-276).
#pragma once
#include <map>
#include <string>
#include <vector>

namespace lashhost {

std::string json_escape(const std::string &s);
std::string json_pretty_string_array(const std::vector<std::string> &items);
std::string json_pretty_string_object(const std::map<std::string, std::string> &kv);   // std::map == sorted keys
// minimal readers for what `lash dist` needs back (main.rs:362-401)
bool json_parse_string_array(const std::string &text, std::vector<std::string> &out);
bool json_parse_string_object(const std::string &text, std::map<std::string, std::string> &out);

}  // namespace lashhost
