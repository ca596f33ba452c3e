"""Hand-written SVG documents for the front-end tests: (name, document, width).  oracle/gen_golden.py --only svg runs the
reference's loader on exactly these strings and stores the scenes it builds (tests/golden/svg_kat.npz)."""

CASES = [
    ("basic_shapes", """<svg xmlns="http://www.w3.org/2000/svg" width="200" height="120" viewBox="0 0 100 60">
  <rect x="5" y="5" width="40" height="20" fill="#369"/>
  <rect x="50" y="5" width="40" height="20" rx="4" ry="7.5" fill="rgb(200, 30, 60)" stroke="black" stroke-width="1.5"/>
  <circle cx="20" cy="42" r="11" fill="orange" fill-opacity="0.5"/>
  <ellipse cx="55" cy="42" rx="14" ry="7" fill="none" stroke="#0a0" stroke-width="2" stroke-linejoin="round"/>
  <line x1="75" y1="32" x2="95" y2="55" stroke="navy" stroke-linecap="round" stroke-width="3"/>
  <polyline points="70,58 80,40 90,58" fill="none" stroke="rgb(10%, 50%, 90%)"/>
  <polygon points="2,58 12,30 22,58" fill="#fc08"/>
</svg>""", None),
    ("groups_transforms_style", """<svg xmlns="http://www.w3.org/2000/svg" viewBox="-10 -10 120 90">
  <g transform="translate(10,5) rotate(15 30 30)" style="fill: teal; stroke: #222; stroke-width: 0.8" opacity="0.7">
    <path d="M0,0 L40,0 L40,30 Q20,50 0,30 Z"/>
    <g transform="matrix(0.5 0.1 -0.2 0.6 40 10)" fill="#c03">
      <path d="M0,0 C10,-20 30,-20 40,0 S70,20 80,0 v25 h-80 z" fill-rule="evenodd" stroke="none"/>
      <path d="M10,30 a10,6 30 1 0 20,5 z" style="fill:none;stroke-opacity:0.4"/>
    </g>
  </g>
  <g transform="skewX(20) scale(0.5, 0.8)"><rect width="30" height="30" x="100" y="60"/></g>
</svg>""", 300),
    ("gradients", """<svg xmlns="http://www.w3.org/2000/svg" width="160" height="100">
  <defs>
    <linearGradient id="lin" x1="0" y1="0" x2="1" y2="1" spreadMethod="reflect">
      <stop offset="0" stop-color="#f00"/><stop offset="50%" stop-color="gold" stop-opacity="0.6"/>
      <stop offset="1" style="stop-color: rgb(0,0,255); stop-opacity: 1"/>
    </linearGradient>
    <radialGradient id="rad" gradientUnits="userSpaceOnUse" cx="110" cy="50" r="40" fx="95" fy="40"
                    gradientTransform="rotate(20 110 50) scale(1 0.7)" color-interpolation="linearRGB">
      <stop offset="0.1" stop-color="white"/><stop offset="0.9" stop-color="#204"/>
    </radialGradient>
    <linearGradient id="one"><stop offset="0.3" stop-color="lime"/></linearGradient>
    <linearGradient id="empty"/>
  </defs>
  <rect x="10" y="10" width="60" height="80" fill="url(#lin)"/>
  <circle cx="110" cy="50" r="40" fill="url(#rad)" stroke="url(#lin)" stroke-width="4"/>
  <rect x="0" y="0" width="10" height="10" fill="url(#one)"/>
  <rect x="150" y="90" width="10" height="10" fill="url(#empty)" stroke="url(#missing)"/>
</svg>""", None),
    ("clip_mask_opacity", """<svg xmlns="http://www.w3.org/2000/svg" width="120" height="120">
  <defs>
    <clipPath id="c1" clip-rule="evenodd" transform="translate(5 5)"><circle cx="50" cy="50" r="40"/><rect x="30" y="30" width="40" height="40"/></clipPath>
    <clipPath id="c2" clipPathUnits="objectBoundingBox"><rect x="0.1" y="0.1" width="0.8" height="0.5"/></clipPath>
    <mask id="m1"><rect x="0" y="0" width="120" height="60" fill="white"/><circle cx="60" cy="60" r="30" fill="#888"/></mask>
  </defs>
  <g clip-path="url(#c1)" opacity="0.9"><rect width="120" height="120" fill="tomato"/><path d="M0,0 L120,120" stroke="black" stroke-width="6"/></g>
  <rect x="20" y="70" width="80" height="40" fill="steelblue" clip-path="url(#c2)" transform="rotate(-5)"/>
  <g mask="url(#m1)"><rect x="10" y="10" width="100" height="100" fill="seagreen"/></g>
</svg>""", 240),
    ("filters", """<svg xmlns="http://www.w3.org/2000/svg" width="100" height="100">
  <defs>
    <filter id="shadow"><feGaussianBlur in="SourceAlpha" stdDeviation="2 3" result="b"/><feOffset in="b" dx="3" dy="2" result="o"/>
      <feMerge><feMergeNode in="o"/><feMergeNode in="SourceGraphic"/></feMerge></filter>
    <filter id="ops"><feColorMatrix type="luminanceToAlpha" result="l"/><feMorphology operator="dilate" radius="1.5" in="l" result="d"/>
      <feComposite in="SourceGraphic" in2="d" operator="arithmetic" k1="0.1" k2="0.8" k3="0.3" k4="0"/>
      <feComposite in2="SourceAlpha" operator="atop"/><feBlend in2="SourceGraphic" mode="multiply"/>
      <feColorMatrix values="1 0 0 0 0  0 1 0 0 0  0 0 1 0 0  0 0 0 0.5 0"/></filter>
    <filter id="hue"><feColorMatrix type="hueRotate" values="75"/><feColorMatrix type="saturate" values="0.35"/>
      <feColorMatrix type="hueRotate"/><feColorMatrix type="saturate"/><feColorMatrix type="sepia"/></filter>
  </defs>
  <path d="M60,70 h35 v25 h-35 z M70,75 v15 h15 z" fill="#3c9" stroke="#b40" filter="url(#hue)"/>
  <circle cx="40" cy="40" r="25" fill="crimson" filter="url(#shadow)"/>
  <g filter="url(#ops)" transform="translate(20 20)"><rect width="50" height="50" fill="#08f"/></g>
</svg>""", None),
    ("nested_svg_use", """<svg xmlns="http://www.w3.org/2000/svg" xmlns:xlink="http://www.w3.org/1999/xlink" viewBox="0 0 200 100">
  <defs><g id="leaf"><path d="M0,0 C5,-10 15,-10 20,0 C15,10 5,10 0,0 z" fill="green"/></g></defs>
  <svg x="10" y="10" width="80" height="80" viewBox="0 0 40 20"><rect width="40" height="20" fill="#eee"/><circle cx="20" cy="10" r="12" fill="purple"/></svg>
  <svg x="110" y="5" width="60" height="40"><rect x="-10" y="-10" width="100" height="100" fill="khaki"/></svg>
  <use xlink:href="#leaf" x="120" y="70"/>
  <use href="#leaf" transform="translate(150 60) scale(1.5)"/>
  <title>ignored</title><text x="5" y="95">skipped</text>
</svg>""", 400),
    ("text_svg_font", """<svg xmlns="http://www.w3.org/2000/svg" width="300" height="120">
  <defs>
    <font id="blocky" horiz-adv-x="600">
      <font-face font-family="Blocky" units-per-em="1000" ascent="800" descent="-200" font-weight="bold"/>
      <missing-glyph horiz-adv-x="500" d="M50,0 H450 V700 H50 z M120,70 V630 H380 V70 z"/>
      <glyph unicode="A" glyph-name="A" d="M0,0 L250,700 L500,0 H400 L340,180 H160 L100,0 z M190,270 H310 L250,450 z"/>
      <glyph unicode="V" glyph-name="V" horiz-adv-x="560" d="M0,700 H100 L250,150 L400,700 H500 L250,0 z"/>
      <glyph unicode="o" glyph-name="o" horiz-adv-x="520" d="M250,0 C80,0 40,120 40,250 S80,500 250,500 S460,380 460,250 S420,0 250,0 z"/>
      <glyph unicode="f" glyph-name="f" horiz-adv-x="300" d="M100,0 V500 Q100,700 280,700 V620 Q180,620 180,500 V0 z"/>
      <glyph unicode="i" glyph-name="i" horiz-adv-x="260" d="M90,0 V500 H170 V0 z M90,600 V700 H170 V600 z"/>
      <glyph unicode="fi" glyph-name="fi" horiz-adv-x="540" d="M100,0 V500 Q100,700 280,700 H450 V0 H370 V620 H280 Q180,620 180,500 V0 z"/>
      <glyph unicode=" " glyph-name="space" horiz-adv-x="280" d=""/>
      <glyph glyph-name="nocode" d="M0,0 H10 V10 z"/>
      <hkern u1="A" u2="V" k="120"/>
      <hkern g1="V,nocode" g2="A,o" k="90"/>
      <hkern u1="o" u2="f"/>
    </font>
  </defs>
  <text x="10" y="50" font-family="Blocky" font-size="40" fill="#234">AVA  fio
    <tspan dy="30" fill="crimson" font-size="20">  Vo?</tspan> of<tspan x="200" dx="-5">A</tspan>
  </text>
  <text x="150" y="100" font-family="blocky" font-size="16" text-anchor="middle" stroke="black" stroke-width="0.5">Ao fi</text>
  <text x="290" y="115" font-family="Blocky" font-size="12" text-anchor="end" fill="none" stroke="#00f">oV</text>
  <text x="5" y="115" font-family="Unknown Sans" font-size="12">no such font</text>
</svg>""", None),
    ("patterns", """<svg xmlns="http://www.w3.org/2000/svg" width="290" height="160">
  <defs>
    <pattern id="dots" x="2" y="3" width="16" height="12" patternUnits="userSpaceOnUse">
      <rect width="16" height="12" fill="#ffe"/><circle cx="8" cy="6" r="4" fill="#c33"/>
      <path d="M0,0 L4,0 L0,3 z" fill="navy" fill-opacity="0.6"/>
    </pattern>
    <pattern id="tilted" width="20" height="10" patternUnits="userSpaceOnUse" patternTransform="rotate(30) scale(0.8 1.1)">
      <rect width="10" height="10" fill="teal"/><rect x="10" width="10" height="5" fill="gold"/>
    </pattern>
    <pattern id="boxed" width="0.25" height="0.5">
      <rect width="10" height="14" fill="#484"/><circle cx="12" cy="18" r="6" fill="#a6f" stroke="black"/>
    </pattern>
    <pattern id="viewboxed" x="5" y="5" width="24" height="24" patternUnits="userSpaceOnUse" viewBox="0 0 4 4">
      <rect width="2" height="2" fill="black"/><rect x="2" y="2" width="2" height="2" fill="#888"/>
    </pattern>
    <pattern id="content_bbox" width="0.5" height="0.5" patternContentUnits="objectBoundingBox">
      <rect width="0.25" height="0.5" fill="crimson"/><circle cx="0.375" cy="0.25" r="0.1" fill="blue"/>
    </pattern>
    <pattern id="nosize"><rect width="5" height="5"/></pattern>
  </defs>
  <rect x="70" y="5" width="90" height="60" fill="url(#dots)" stroke="black"/>
  <ellipse cx="230" cy="35" rx="50" ry="28" fill="url(#tilted)"/>
  <path d="M10,80 h80 v60 q-40,25 -80,0 z" fill="url(#boxed)"/>
  <g transform="translate(110 75) rotate(-10)"><rect width="70" height="50" fill="url(#viewboxed)"/></g>
  <circle cx="30" cy="30" r="26" fill="url(#content_bbox)" stroke="url(#dots)" stroke-width="5"/>
  <circle cx="230" cy="120" r="30" fill="url(#dots)" fill-opacity="0.5"/>
  <rect x="100" y="140" width="20" height="15" fill="url(#nosize)"/>
</svg>""", None),
    # the reference hands the raw ``color`` string on as the paint (S:3147-3148): loadable, not renderable
    ("current_color", """<svg xmlns="http://www.w3.org/2000/svg" width="20" height="20">
  <g color="#c03"><rect width="10" height="10" fill="currentColor"/><path d="M2,12 h16" stroke="currentColor"/></g>
</svg>""", None),
    ("empty", """<svg xmlns="http://www.w3.org/2000/svg" width="10" height="10"><defs/></svg>""", None),
]
