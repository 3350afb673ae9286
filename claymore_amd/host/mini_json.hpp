// mini_json.hpp — a small recursive-descent JSON reader for scene files (the reference uses rapidjson,
// Projects/GMPM/gmpm.cu:60-165; only objects, arrays, numbers, strings, booleans and null are needed).
#pragma once
#include <cctype>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace mj {
struct Value;
using ValuePtr = std::shared_ptr<Value>;
struct Value {
	enum Type { Null, Bool, Number, String, Array, Object } type = Null;
	bool b		 = false;
	double num	 = 0.0;
	std::string str;
	std::vector<ValuePtr> arr;
	std::map<std::string, ValuePtr> obj;
	bool has(const std::string& k) const { return type == Object && obj.count(k) > 0; }
	const Value& operator[](const std::string& k) const {
		auto it = obj.find(k);
		if(type != Object || it == obj.end()) throw std::runtime_error("missing key: " + k);
		return *it->second;
	}
	const Value& operator[](size_t i) const {
		if(type != Array || i >= arr.size()) throw std::runtime_error("bad array index");
		return *arr[i];
	}
	double number() const {
		if(type != Number) throw std::runtime_error("not a number");
		return num;
	}
	const std::string& string() const {
		if(type != String) throw std::runtime_error("not a string");
		return str;
	}
};

class Parser {
	const std::string& s;
	size_t i = 0;
	void ws() {
		while(i < s.size() && std::isspace((unsigned char) s[i])) ++i;
	}
	[[noreturn]] void fail(const char* m) { throw std::runtime_error(std::string("json: ") + m + " at offset " + std::to_string(i)); }
	ValuePtr value() {
		ws();
		if(i >= s.size()) fail("unexpected end");
		auto v = std::make_shared<Value>();
		const char c = s[i];
		if(c == '{') {
			v->type = Value::Object;
			++i;
			ws();
			if(i < s.size() && s[i] == '}') {
				++i;
				return v;
			}
			for(;;) {
				ws();
				ValuePtr k = value();
				if(k->type != Value::String) fail("object key must be a string");
				ws();
				if(i >= s.size() || s[i] != ':') fail("expected ':'");
				++i;
				v->obj[k->str] = value();
				ws();
				if(i < s.size() && s[i] == ',') {
					++i;
					continue;
				}
				if(i < s.size() && s[i] == '}') {
					++i;
					break;
				}
				fail("expected ',' or '}'");
			}
		} else if(c == '[') {
			v->type = Value::Array;
			++i;
			ws();
			if(i < s.size() && s[i] == ']') {
				++i;
				return v;
			}
			for(;;) {
				v->arr.push_back(value());
				ws();
				if(i < s.size() && s[i] == ',') {
					++i;
					continue;
				}
				if(i < s.size() && s[i] == ']') {
					++i;
					break;
				}
				fail("expected ',' or ']'");
			}
		} else if(c == '"') {
			v->type = Value::String;
			++i;
			while(i < s.size() && s[i] != '"') {
				if(s[i] == '\\' && i + 1 < s.size()) {
					++i;
					switch(s[i]) {
						case 'n': v->str += '\n'; break;
						case 't': v->str += '\t'; break;
						default: v->str += s[i]; break;
					}
				} else {
					v->str += s[i];
				}
				++i;
			}
			if(i >= s.size()) fail("unterminated string");
			++i;
		} else if(s.compare(i, 4, "true") == 0) {
			v->type = Value::Bool;
			v->b	= true;
			i += 4;
		} else if(s.compare(i, 5, "false") == 0) {
			v->type = Value::Bool;
			i += 5;
		} else if(s.compare(i, 4, "null") == 0) {
			i += 4;
		} else {
			char* end = nullptr;
			v->num	  = std::strtod(s.c_str() + i, &end);
			if(end == s.c_str() + i) fail("unexpected character");
			v->type = Value::Number;
			i		= (size_t) (end - s.c_str());
		}
		return v;
	}

   public:
	explicit Parser(const std::string& text) : s(text) {}
	ValuePtr parse() {
		ValuePtr v = value();
		ws();
		if(i != s.size()) fail("trailing characters");
		return v;
	}
};
inline ValuePtr parse(const std::string& text) {
	return Parser(text).parse();
}
}// namespace mj
