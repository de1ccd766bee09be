// xml.hpp -- minimal XML reader for SKIRT parameter files (.ski).
//
// A ski file is plain XML 1.0: a declaration, optional comments, and a tree of elements whose attributes carry
// the property values (reference reader: SMILE/fundamentals/XmlReader.cpp, driven by
// SMILE/serialize/XmlHierarchyCreator.cpp).  This reader supports exactly that subset: elements, attributes
// (single or double quotes), comments, the <?xml ...?> declaration, the five predefined entities and numeric
// character references.  No DTDs, namespaces, CDATA or mixed content.
#ifndef SKH_XML_HPP
#define SKH_XML_HPP

#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace skh
{
    struct XmlElement
    {
        std::string name;
        std::vector<std::pair<std::string, std::string>> attributes;
        std::vector<std::unique_ptr<XmlElement>> children;
        int line{0};

        bool has(const std::string& key) const
        {
            for (auto& a : attributes)
                if (a.first == key) return true;
            return false;
        }
        const std::string& attr(const std::string& key) const
        {
            for (auto& a : attributes)
                if (a.first == key) return a.second;
            throw std::runtime_error("ski: element <" + name + "> (line " + std::to_string(line)
                                     + ") has no attribute '" + key + "'");
        }
        std::string attr(const std::string& key, const std::string& fallback) const
        {
            for (auto& a : attributes)
                if (a.first == key) return a.second;
            return fallback;
        }
        // the property element with the given name (e.g. <geometry type="Geometry">), or nullptr
        const XmlElement* child(const std::string& childName) const
        {
            for (auto& c : children)
                if (c->name == childName) return c.get();
            return nullptr;
        }
        // the single item inside a property element, or nullptr if the property is absent/empty
        const XmlElement* item(const std::string& propertyName) const
        {
            const XmlElement* p = child(propertyName);
            if (!p || p->children.empty()) return nullptr;
            return p->children.front().get();
        }
        // all items inside a list property element
        std::vector<const XmlElement*> items(const std::string& propertyName) const
        {
            std::vector<const XmlElement*> result;
            const XmlElement* p = child(propertyName);
            if (p)
                for (auto& c : p->children) result.push_back(c.get());
            return result;
        }
    };

    class XmlParser
    {
    public:
        explicit XmlParser(const std::string& text) : _s(text) {}

        std::unique_ptr<XmlElement> parseDocument()
        {
            skipMisc();
            auto root = parseElement();
            skipMisc();
            if (_p != _s.size()) fail("unexpected content after the root element");
            return root;
        }

    private:
        const std::string& _s;
        size_t _p{0};
        int _line{1};

        [[noreturn]] void fail(const std::string& msg) const
        {
            throw std::runtime_error("ski: XML error at line " + std::to_string(_line) + ": " + msg);
        }
        bool startsWith(const char* lit) const { return _s.compare(_p, std::char_traits<char>::length(lit), lit) == 0; }
        void advance(size_t n)
        {
            for (size_t i = 0; i < n && _p < _s.size(); ++i, ++_p)
                if (_s[_p] == '\n') ++_line;
        }
        void skipSpace()
        {
            while (_p < _s.size() && (_s[_p] == ' ' || _s[_p] == '\t' || _s[_p] == '\r' || _s[_p] == '\n')) advance(1);
        }
        void skipUntil(const char* lit)
        {
            size_t q = _s.find(lit, _p);
            if (q == std::string::npos) fail(std::string("missing '") + lit + "'");
            advance(q - _p + std::char_traits<char>::length(lit));
        }
        // whitespace, comments, processing instructions, doctype
        void skipMisc()
        {
            while (true)
            {
                skipSpace();
                if (startsWith("<!--"))
                    skipUntil("-->");
                else if (startsWith("<?"))
                    skipUntil("?>");
                else if (startsWith("<!DOCTYPE"))
                    skipUntil(">");
                else
                    break;
            }
        }
        static bool nameChar(char c)
        {
            return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_' || c == '-'
                   || c == '.' || c == ':';
        }
        std::string parseName()
        {
            size_t b = _p;
            while (_p < _s.size() && nameChar(_s[_p])) ++_p;
            if (_p == b) fail("expected a name");
            return _s.substr(b, _p - b);
        }
        std::string decode(const std::string& raw)
        {
            std::string out;
            out.reserve(raw.size());
            for (size_t i = 0; i < raw.size(); ++i)
            {
                if (raw[i] != '&')
                {
                    out += raw[i];
                    continue;
                }
                size_t e = raw.find(';', i);
                if (e == std::string::npos) fail("unterminated entity");
                std::string ent = raw.substr(i + 1, e - i - 1);
                if (ent == "lt")
                    out += '<';
                else if (ent == "gt")
                    out += '>';
                else if (ent == "amp")
                    out += '&';
                else if (ent == "quot")
                    out += '"';
                else if (ent == "apos")
                    out += '\'';
                else if (!ent.empty() && ent[0] == '#')
                {
                    unsigned long code =
                        (ent.size() > 1 && ent[1] == 'x') ? std::stoul(ent.substr(2), nullptr, 16) : std::stoul(ent.substr(1));
                    // encode as UTF-8
                    if (code < 0x80)
                        out += static_cast<char>(code);
                    else if (code < 0x800)
                    {
                        out += static_cast<char>(0xC0 | (code >> 6));
                        out += static_cast<char>(0x80 | (code & 0x3F));
                    }
                    else
                    {
                        out += static_cast<char>(0xE0 | (code >> 12));
                        out += static_cast<char>(0x80 | ((code >> 6) & 0x3F));
                        out += static_cast<char>(0x80 | (code & 0x3F));
                    }
                }
                else
                    fail("unknown entity &" + ent + ";");
                i = e;
            }
            return out;
        }
        std::unique_ptr<XmlElement> parseElement()
        {
            if (_p >= _s.size() || _s[_p] != '<') fail("expected '<'");
            advance(1);
            auto elem = std::make_unique<XmlElement>();
            elem->line = _line;
            elem->name = parseName();
            while (true)
            {
                skipSpace();
                if (_p >= _s.size()) fail("unterminated start tag");
                if (startsWith("/>"))
                {
                    advance(2);
                    return elem;
                }
                if (_s[_p] == '>')
                {
                    advance(1);
                    break;
                }
                std::string key = parseName();
                skipSpace();
                if (_p >= _s.size() || _s[_p] != '=') fail("expected '=' after attribute name");
                advance(1);
                skipSpace();
                if (_p >= _s.size() || (_s[_p] != '"' && _s[_p] != '\'')) fail("expected a quoted attribute value");
                char quote = _s[_p];
                advance(1);
                size_t e = _s.find(quote, _p);
                if (e == std::string::npos) fail("unterminated attribute value");
                std::string raw = _s.substr(_p, e - _p);
                advance(e - _p + 1);
                elem->attributes.emplace_back(key, decode(raw));
            }
            // content: child elements only (text is ignored apart from whitespace)
            while (true)
            {
                skipMisc();
                if (_p >= _s.size()) fail("unterminated element <" + elem->name + ">");
                if (startsWith("</"))
                {
                    advance(2);
                    std::string closing = parseName();
                    if (closing != elem->name) fail("mismatched closing tag </" + closing + "> for <" + elem->name + ">");
                    skipSpace();
                    if (_p >= _s.size() || _s[_p] != '>') fail("expected '>'");
                    advance(1);
                    return elem;
                }
                if (_s[_p] == '<')
                    elem->children.push_back(parseElement());
                else
                    fail("unexpected character data inside <" + elem->name + ">");
            }
        }
    };
}

#endif
