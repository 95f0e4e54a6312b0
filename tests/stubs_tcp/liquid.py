"""Stand-in for python-liquid (prompt templating of the generation side, out of scope; imported by ragroute/llm_message.py)."""


class Template:
    def __init__(self, text):
        self.text = text

    def render(self, **kw):
        return self.text
