"""Stand-in for the `ollama` client: the LLM is out of scope and disabled (`--disable-llm`); the front-end imports the names."""


class ChatResponse(dict):
    pass


class AsyncClient:
    async def chat(self, *a, **k):
        raise RuntimeError("ollama is not available: run with --disable-llm")
