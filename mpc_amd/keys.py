"""PRgenerator keys of the engines (mpc_engine/ABY3Engine.hpp:10-15; the same in the other two engine headers)."""
KEY01, KEY12, KEY20, KEYS = 0xAB01, 0xBC12, 0xCA20, 0xF0
