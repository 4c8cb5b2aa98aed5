// ConfigParser -- the configuration reader of ROFT-tracker (reference: src/roft/include/ConfigParser.h:21-118,
// src/roft/src/ConfigParser.cpp:12-169): a libconfig file named by `--from <path>` (or the constructor argument), every
// setting of which can be overridden on the command line as `--group::sub::key value` -- booleans as true / false, arrays as
// "x_1, ..., x_n" with as many values as the file gives the setting -- and is read back with conf("group.sub.key", variable).
// The reference builds this on libconfig++ and TCLAP; neither is installed where this repository is built, so this is a
// reader of the same files and the same command lines written from scratch (roft_amd/config.py is its Python twin).
// Supported syntax: `name = value;` / `name : value;`, groups `{ ... }`, arrays `[ ... ]`, lists `( ... )`, strings, integers
// (also with an L suffix), floats, true / false, comments `#`, `//`, `/* */`.
#pragma once

#include <cctype>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include <Eigen/Dense>

class ConfigParser {
public:
    ConfigParser(const int& argc, char** argv, const std::string& file_path = "")
    {
        std::string cfg_path = file_path;
        for (int i = 1; i + 1 < argc; ++i)
            if (std::string(argv[i]) == "--from") cfg_path = argv[i + 1];
        if (cfg_path.empty())
            throw std::runtime_error("ConfigParser::ctor. Please provide a valid configuration file using --from <path_to_cfg_file>");
        std::ifstream in(cfg_path);
        if (!in) throw std::runtime_error("ConfigParser::ctor. I/O error while reading " + cfg_path + ".");
        std::stringstream buffer;
        buffer << in.rdbuf();
        text_ = buffer.str();
        file_ = cfg_path;
        root_.kind = Node::Group;
        parse_group(root_, '\0');
        skip_space();
        if (pos_ != text_.size()) parse_error("syntax error");
        // command line: every argument must be `--from <path>` or `--<setting> <value>` of a setting of the file
        for (int i = 1; i < argc; ++i) {
            const std::string a = argv[i];
            if (a.rfind("--", 0) != 0 || i + 1 >= argc) throw std::runtime_error("ConfigParser::ctor. Error: cannot parse the argument " + a);
            const std::string value = argv[++i];
            if (a == "--from") continue;
            Node* n = find(a.substr(2));
            if (!n || n->kind == Node::Group) throw std::runtime_error("ConfigParser::ctor. Error: " + a + " is not a setting of " + cfg_path);
            override_setting(*n, a, value);
        }
    }

    void operator()(const std::string& path, double& value) { if (const Node* n = find(path)) { if (n->kind == Node::Float || n->kind == Node::Int) value = n->number; } }
    void operator()(const std::string& path, int& value) { if (const Node* n = find(path)) { if (n->kind == Node::Int) value = (int)n->number; } }
    void operator()(const std::string& path, bool& value) { if (const Node* n = find(path)) { if (n->kind == Node::Bool) value = n->number != 0.0; } }
    void operator()(const std::string& path, std::string& value) { if (const Node* n = find(path)) { if (n->kind == Node::String) value = n->text; } }
    template <class T>
    void operator()(const std::string& path, std::vector<T>& array)
    {
        const Node* n = find(path);
        if (!n) throw std::runtime_error("ConfigParser::operator(). Error: cannot find the setting with name " + dotted(path));
        if (n->kind != Node::Array) throw std::runtime_error("ConfigParser::operator(). Error: cannot find an array setting with name " + dotted(path));
        for (const auto& item : n->items) array.push_back((T)item->number);
    }
    void operator()(const std::string& path, Eigen::VectorXd& array)
    {
        std::vector<double> v;
        operator()(path, v);
        array.resize(v.size());
        for (std::size_t i = 0; i < v.size(); ++i) array(i) = v[i];
    }

private:
    struct Node {
        enum Kind { Group, Array, Int, Float, Bool, String } kind = Group;
        double number = 0.0;
        std::string text;
        std::vector<std::pair<std::string, std::unique_ptr<Node>>> members;   // Group
        std::vector<std::unique_ptr<Node>> items;                             // Array / list
    };

    static std::string dotted(const std::string& path)
    {
        std::string out;
        for (std::size_t i = 0; i < path.size(); ++i) {
            if (path.compare(i, 2, "::") == 0) { out += '.'; ++i; }
            else out += path[i];
        }
        return out;
    }
    Node* find(const std::string& path)
    {
        Node* n = &root_;
        std::stringstream parts(dotted(path));
        std::string part;
        while (std::getline(parts, part, '.')) {
            if (n->kind != Node::Group) return nullptr;
            Node* next = nullptr;
            for (auto& m : n->members)
                if (m.first == part) next = m.second.get();
            if (!next) return nullptr;
            n = next;
        }
        return n;
    }
    void override_setting(Node& n, const std::string& arg, const std::string& raw)
    {
        auto number = [&](const std::string& s, bool integer) {
            std::size_t used = 0;
            double v = 0.0;
            try { v = integer ? (double)std::stol(s, &used) : std::stod(s, &used); } catch (const std::exception&) { used = 0; }
            while (used < s.size() && std::isspace((unsigned char)s[used])) ++used;
            if (used != s.size() || s.empty()) throw std::runtime_error("ConfigParser::ctor. Error: cannot parse the value of " + arg + ": " + raw);
            return v;
        };
        switch (n.kind) {
        case Node::Bool:
            if (raw != "true" && raw != "false") throw std::runtime_error("ConfigParser::ctor. Error: " + arg + " takes true or false");
            n.number = raw == "true";
            break;
        case Node::Int: n.number = number(raw, true); break;
        case Node::Float: n.number = number(raw, false); break;
        case Node::String: n.text = raw; break;
        case Node::Array: {
            const bool integer = !n.items.empty() && n.items[0]->kind == Node::Int;
            std::vector<double> values;
            std::string item;
            std::stringstream ss(raw);
            while (std::getline(ss, item, ',')) {
                std::size_t a = 0, b = item.size();
                while (a < b && std::isspace((unsigned char)item[a])) ++a;
                while (b > a && std::isspace((unsigned char)item[b - 1])) --b;
                values.push_back(number(item.substr(a, b - a), integer));
            }
            if (values.size() != n.items.size())
                throw std::runtime_error("ConfigParser::ctor. Error: " + arg + " takes \"x_1, ..., x_" + std::to_string(n.items.size()) + "\"");
            for (std::size_t i = 0; i < values.size(); ++i) n.items[i]->number = values[i];
            break;
        }
        default: break;
        }
    }

    // ---- parser
    [[noreturn]] void parse_error(const std::string& what)
    {
        std::size_t line = 1;
        for (std::size_t i = 0; i < pos_ && i < text_.size(); ++i) line += text_[i] == '\n';
        throw std::runtime_error("ConfigParser::ctor. Parse error at " + file_ + ":" + std::to_string(line) + " - " + what);
    }
    void skip_space()
    {
        for (;;) {
            while (pos_ < text_.size() && std::isspace((unsigned char)text_[pos_])) ++pos_;
            if (pos_ < text_.size() && (text_[pos_] == '#' || text_.compare(pos_, 2, "//") == 0)) {
                while (pos_ < text_.size() && text_[pos_] != '\n') ++pos_;
            } else if (text_.compare(pos_, 2, "/*") == 0) {
                const std::size_t end = text_.find("*/", pos_ + 2);
                if (end == std::string::npos) parse_error("unterminated comment");
                pos_ = end + 2;
            } else return;
        }
    }
    char peek() { skip_space(); return pos_ < text_.size() ? text_[pos_] : '\0'; }
    void parse_group(Node& g, char closing)
    {
        while (peek() != closing) {
            if (peek() == '\0') parse_error("unexpected end of file");
            std::size_t b = pos_;
            while (pos_ < text_.size() && (std::isalnum((unsigned char)text_[pos_]) || text_[pos_] == '_' || text_[pos_] == '-' || text_[pos_] == '*')) ++pos_;
            if (pos_ == b) parse_error("syntax error");
            const std::string name = text_.substr(b, pos_ - b);
            const char op = peek();
            if (op != '=' && op != ':') parse_error("expected '=' or ':' after " + name);
            ++pos_;
            g.members.emplace_back(name, parse_value());
            while (peek() == ';' || peek() == ',') ++pos_;
        }
    }
    std::unique_ptr<Node> parse_value()
    {
        auto n = std::make_unique<Node>();
        const char c = peek();
        if (c == '{') {
            ++pos_;
            n->kind = Node::Group;
            parse_group(*n, '}');
            ++pos_;
        } else if (c == '[' || c == '(') {
            const char closing = c == '[' ? ']' : ')';
            ++pos_;
            n->kind = Node::Array;
            while (peek() != closing) {
                if (peek() == '\0') parse_error("unexpected end of file");
                n->items.push_back(parse_value());
                if (peek() == ',') ++pos_;
            }
            ++pos_;
        } else if (c == '"') {
            n->kind = Node::String;
            // (adjacent string literals are concatenated)
            while (peek() == '"') {
                ++pos_;
                while (pos_ < text_.size() && text_[pos_] != '"') {
                    if (text_[pos_] == '\\' && pos_ + 1 < text_.size()) {
                        const char e = text_[++pos_];
                        n->text += e == 'n' ? '\n' : (e == 't' ? '\t' : (e == 'r' ? '\r' : e));
                    } else n->text += text_[pos_];
                    ++pos_;
                }
                if (pos_ >= text_.size()) parse_error("unterminated string");
                ++pos_;
            }
        } else {
            std::size_t b = pos_;
            while (pos_ < text_.size() && (std::isalnum((unsigned char)text_[pos_]) || text_[pos_] == '.' || text_[pos_] == '+' || text_[pos_] == '-')) ++pos_;
            std::string tok = text_.substr(b, pos_ - b);
            if (tok.empty()) parse_error("syntax error");
            std::string low;
            for (char ch : tok) low += (char)std::tolower((unsigned char)ch);
            if (low == "true" || low == "false") { n->kind = Node::Bool; n->number = low == "true"; return n; }
            while (!tok.empty() && tok.back() == 'L') tok.pop_back();
            const bool hex = low.rfind("0x", 0) == 0;
            const bool integer = hex || tok.find_first_of(".eE") == std::string::npos;
            std::size_t used = 0;
            try {
                if (integer) n->number = (double)std::stoll(tok, &used, hex ? 16 : 10);
                else n->number = std::stod(tok, &used);
            } catch (const std::exception&) { used = 0; }
            if (used != tok.size()) parse_error("syntax error");
            n->kind = integer ? Node::Int : Node::Float;
        }
        return n;
    }

    std::string text_, file_;
    std::size_t pos_ = 0;
    Node root_;
};
